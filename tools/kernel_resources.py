"""Register / scratch / LDS budget of every kernel in one csrc/*.hip file, from hipcc's own resource-usage remarks
(no GPU needed: the compiler's view).  Usage:  python tools/kernel_resources.py gemm_bf16.hip [substring ...]

A hand-scheduled kernel that starts spilling after an edit shows up here (ScratchSize > 0) long before it shows up as a
slow launch on the GPU box."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "traffic-context-augmented-vehicle-trajectory-prediction-framework-using-multimodal-llm_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-std=c++17"]


def demangle(names):
    import shutil

    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if tool is None:
        return names
    out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else names


def main():
    src = sys.argv[1]
    pats = sys.argv[2:]
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", "/dev/null",
                                                           "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = demangle([r_["name"] for r_ in rows])
    print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS':>7} {'occ':>4}  kernel")
    for r_, n in zip(rows, names):
        if pats and not all(p in n for p in pats):
            continue
        print(f"{r_.get('VGPRs', '?'):>5} {r_.get('AGPRs', '?'):>5} {r_.get('SGPRs', '?'):>5} "
              f"{r_.get('ScratchSize [bytes/lane]', '?'):>8} {r_.get('LDS Size [bytes/block]', '?'):>7} "
              f"{r_.get('Occupancy [waves/SIMD]', '?'):>4}  {n[:150]}")


if __name__ == "__main__":
    main()
