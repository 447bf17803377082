"""Find global loads that the compiler serialised: a vector-memory load followed -- before the next load -- by `s_waitcnt vmcnt(0)`.

A load under a lane-varying condition (`i < n ? p[i] : 0`, `if (ok) v = p[i]`) in an unrolled loop becomes a branch per load, and the
memory counter is drained at every join: N loads that the source shows side by side go out as N dependent round trips (0.5-2 us each).
Round 5 found the single-workgroup top-K kernels (40 loads, a third of the kernel), the GroupNorm partial sums (32), the pooled taps of
the fused eSE pass (18) and the proposal selection (16) that way; the fix is a clamped index + an unconditional load + a select.
It is a screen, not a verdict: the register-staged igemm kernel is flagged too, and un-predicating its loads measured no gain
(profiles/r5/igemm_unpredicated_loads_ab.txt) -- its K step is bound by the LDS round trip + barrier, not by the global loads.

  python tools/scan_serial_loads.py [file.hip ...]      # default: every csrc/*.hip except the two igemm units (4 minutes each)
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from far3d_amd import build as fbuild  # noqa: E402

LOAD = re.compile(r"(global_load|buffer_load|flat_load)")


def asm_of(src):
    out = os.path.join("/tmp/far3d_scan_asm", os.path.basename(src)[:-4] + ".s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run([fbuild.HIPCC] + fbuild.FLAGS + ["-S", "--cuda-device-only", os.path.join(fbuild.CSRC, src), "-o", out], check=True,
                   capture_output=True)
    return out


def scan(path, threshold=4):
    txt = open(path).read()
    rows = []
    for m in re.finditer(r"^(_Z\w+):\s*;.*?\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]
        n_load = sum(1 for ln in lines if LOAD.match(ln) and "lds" not in ln)
        serial = 0
        for i, ln in enumerate(lines):
            if LOAD.match(ln) and "lds" not in ln:
                for nx in lines[i + 1:i + 8]:
                    if LOAD.match(nx):
                        break
                    if nx.startswith("s_waitcnt") and "vmcnt(0)" in nx:
                        serial += 1
                        break
        if serial >= threshold:
            rows.append((os.path.basename(path), name, n_load, serial))
    return rows


def main():
    srcs = sys.argv[1:] or [s for s in fbuild._sources() if not s.startswith("igemm")]
    with ThreadPoolExecutor(max_workers=6) as ex:
        paths = list(ex.map(asm_of, srcs))
    for p in paths:
        for f, name, n_load, serial in scan(p):
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            print("%-14s %-90s loads %4d  load -> vmcnt(0) pairs %3d" % (f, dem[:90], n_load, serial))


if __name__ == "__main__":
    main()
