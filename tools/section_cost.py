#!/usr/bin/env python3
"""Cost of each section of lattice_fused_kernel, measured by repetition (run on a GPU box; patches a COPY of the source).

The kernel source carries `// [section NAME]` ... `// [/section NAME]` comment pairs around idempotent sections.  For each
section this tool builds a variant of libfrenetgpu.so in which the section runs REP times (a `for` loop with a compiler
memory barrier around it: results are unchanged), times the kernel with bench.py under rocprofv3 and reports
(t_REP - t_1) / (REP - 1).  Repetition keeps the control flow and the data intact, which removing a section does not.

    python tools/section_cost.py [--rep 3] [--config 3]
"""
import os
os.environ.setdefault("FP_ALLOW_DIAGNOSTIC_BUILD", "1")  # runs against a library built with EXTRA=-DFP_...
import argparse, csv, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fiss_plus_planner_amd", "csrc", "frenet_lattice_fused.hip")


def patched(text, name, rep):
    a, b = f"    // [section {name}]\n", f"    // [/section {name}]\n"
    return text.replace(a, f'    for (int rep_ = 0; rep_ < {rep}; ++rep_) {{ asm volatile("" ::: "memory");\n').replace(b, "    }\n")


def kernel_us(config):
    out = tempfile.mkdtemp(prefix="sect_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "v", "--", sys.executable,
                    os.path.join(ROOT, "bench.py"), "--config", str(config), "--steps", "20", "--no-latency", "--cpu-seconds", "0"],
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    for root, _, files in os.walk(out):
        for f in files:
            if f.endswith("kernel_stats.csv"):
                for r in csv.DictReader(open(os.path.join(root, f))):
                    if "lattice_fused" in r["Name"]:
                        shutil.rmtree(out, ignore_errors=True)
                        return float(r["AverageNs"]) / 1e3
    raise RuntimeError("lattice_fused_kernel not found in the trace")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rep", type=int, default=3)
    ap.add_argument("--config", type=int, default=3)
    args = ap.parse_args()
    original = open(SRC).read()
    names = re.findall(r"// \[section (\w+)\]", original)
    try:
        base = None
        for name in [None] + names:
            open(SRC, "w").write(original if name is None else patched(original, name, args.rep))
            subprocess.run(["make", "-C", os.path.dirname(SRC)], check=True, stdout=subprocess.DEVNULL)
            t = kernel_us(args.config)
            if name is None:
                base = t
                print(f"kernel {t:8.1f} us")
            else:
                print(f"  {name:8s} {(t - base) / (args.rep - 1):7.1f} us  ({100 * (t - base) / (args.rep - 1) / base:4.1f} %)")
    finally:
        open(SRC, "w").write(original)
        subprocess.run(["make", "-C", os.path.dirname(SRC)], check=True, stdout=subprocess.DEVNULL)


if __name__ == "__main__":
    main()
