"""Static instruction counts of one k_sift_s instantiation (gfx950 ISA), by class and by SOURCE LINE (debug line tables) - the kernel issues at the vector
rate, so what it costs is its vector instruction count.
usage: python tools/sift_isa.py [W NT [extra hipcc flags ...]]   (default 13 9; writes /tmp/isa/sift.s)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from readtape_amd import build as b
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 13
    NT = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    extra = sys.argv[3:]
    os.makedirs("/tmp/isa", exist_ok=True)
    src = "/tmp/isa/sift_one.hip"
    with open(src, "w") as f:
        f.write('#include "rtfe_sift.hip"\nnamespace rtfe { template __global__ void k_sift_s<%d, %d, 6, true>(const SfArgs); }\n' % (W, NT))
    flags = [x for x in b.HIP_FLAGS if x not in ("-shared", "-fPIC")]
    cmd = [b.HIPCC] + flags + extra + ["-gline-tables-only", "-S", "--cuda-device-only", "-o", "/tmp/isa/sift.s", src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    text = open("/tmp/isa/sift.s").read()
    files = {int(m.group(1)): m.group(2) for m in re.finditer(r'\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]+)"', text)}
    i0 = text.index("\n_ZN4rtfe8k_sift_s")
    i1 = text.index("s_endpgm", i0)
    body = text[text.index("\n", i0 + 1): i1]
    tail = text[i1: i1 + 6000]
    cls = collections.Counter()
    per_line = collections.defaultdict(collections.Counter)
    loc = ("?", 0)
    for line in body.splitlines():
        t = line.strip()
        if t.startswith(".loc"):
            p = t.split()
            loc = (os.path.basename(files.get(int(p[1]), "?")), int(p[2]))
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") or re.match(r"^\.?\w+:", t):
            continue
        op = t.split()[0]
        k = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        cls[k] += 1; per_line[loc][k] += 1
        if op in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"): cls[op] += 1
        if op.startswith("s_waitcnt"): cls["s_waitcnt"] += 1
        if op == "s_barrier": cls["s_barrier"] += 1
    print(dict(cls))
    for key in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy"):
        mm = re.search(r";\s*" + key + r":\s*(\d+)", tail)
        if mm: print(key, mm.group(1), end="  ")
    print()
    if os.environ.get("LINES", "1") != "0":
        srcl = open(os.path.join(b.CSRC, "rtfe_sift.hip")).read().splitlines()
        for (fn, ln), c in sorted(per_line.items()):
            if c["valu"] + c["lds"] >= int(os.environ.get("MIN", "4")):
                txt = srcl[ln - 1].strip()[:110] if fn == "rtfe_sift.hip" and 0 < ln <= len(srcl) else ""
                print(f"  {fn}:{ln:4d} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d}  {txt}")


if __name__ == "__main__":
    main()
