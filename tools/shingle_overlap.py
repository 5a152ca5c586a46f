"""8-token shingle overlap of repo files against the reference tree (build container only: needs /root/reference).
Usage: python tools/shingle_overlap.py [files...]   (default: every source file outside oracle/ larger than 2 kB)"""
import os
import re
import sys

REF = os.environ.get("ORB_REFERENCE_ROOT", "/root/reference")
TOK = re.compile(r"[A-Za-z_][A-Za-z0-9_]*|\d+\.?\d*|[^\sA-Za-z0-9_]")


def shingles(path, k=8):
    t = TOK.findall(re.sub(r"//[^\n]*|/\*.*?\*/|#[^\n]*", " ", open(path, errors="ignore").read(), flags=re.S))
    return set(tuple(t[i:i + k]) for i in range(len(t) - k + 1))


def main():
    ref = set()
    for d, _, fs in os.walk(REF):
        for f in fs:
            if f.endswith((".cc", ".cpp", ".h", ".hpp")):
                ref |= shingles(os.path.join(d, f))
    files = sys.argv[1:]
    if not files:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for d, _, fs in os.walk(root):
            if any(x in d for x in ("/.git", "/gpurun_out", "/oracle/_", "/lib", "/__pycache__", "/profiles")):
                continue
            for f in fs:
                p = os.path.join(d, f)
                if f.endswith((".h", ".hip", ".cpp", ".py")) and os.path.getsize(p) > 2000:
                    files.append(p)
    rows = []
    for p in files:
        s = shingles(p)
        if s:
            rows.append((100.0 * len(s & ref) / len(s), p))
    for pct, p in sorted(rows, reverse=True)[:25]:
        print("%5.1f %%  %s" % (pct, p))


if __name__ == "__main__":
    main()
