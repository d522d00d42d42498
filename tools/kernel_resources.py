"""Per-kernel register / scratch / occupancy table for gfx950, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
    python tools/kernel_resources.py [-D MACRO=VALUE ...] > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msdfgen_amd import build as B  # noqa: E402


def main():
    extra = sys.argv[1:]
    cmd = [B.hipcc()] + B.HIPCC_FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", os.path.join(B.CSRC, "msdf_capi.hip"), "-o", "/tmp/_kres.so"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+: +([A-Za-z ]+?)(?: \[bytes/(?:lane|block)\])?: (\S+)", line) or re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        m2 = re.search(r"Function Name: (\S+)", line)
        if m2:
            cur = {"name": m2.group(1)}
            rows.append(cur)
        elif cur is not None:
            m3 = re.search(r"remark: (?:[^ ]+ )?\s*([A-Za-z][A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
            if m3:
                cur[m3.group(1).strip()] = m3.group(2)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("%-58s %5s %5s %9s %9s %8s %5s %7s" % ("kernel", "VGPR", "SGPR", "VGPRspill", "SGPRspill", "scratchB", "occ", "LDS B"))
    for r, n in zip(rows, names):
        n = re.sub(r"^void msdfhip::", "", n)
        n = re.sub(r"\(.*$", "", n)
        print("%-58s %5s %5s %9s %9s %8s %5s %7s" % (n, r.get("VGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
                                                     r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS Size")))


if __name__ == "__main__":
    main()
