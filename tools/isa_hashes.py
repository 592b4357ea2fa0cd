#!/usr/bin/env python3
"""ISA fingerprints of the lane-per-block 8x8 kernels, one kernel per translation unit.

    tools/isa_hashes.py [files]    print  <file> <hash> vgpr <n> spills <v>/<s> scratch <b>  <kernel>  for the given kernel sources
                                   (default: every k_lane8_*.hip / k_cols8_*.hip)
    tools/isa_hashes.py --update   rewrite tests/golden/isa_hashes.txt from the sources as they are
    tools/isa_hashes.py --check    compare with tests/golden/isa_hashes.txt (what tests/test_isa_hashes.py does)

The code hipcc emits for a kernel depended on which OTHER kernels of its translation unit inlined the same instantiation
of search_block (LAB_LOG.md rounds 4 and 5: an added kernel moved untouched ones by 2.5 %); since round 5 every shipped
instantiation has a translation unit of its own, and this file pins what each one compiles to: a change that moves a
hash has to say so by updating the golden file in the same commit.  hipcc cross-compiles gfx950 without a GPU.
"""
import concurrent.futures
import glob
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aero-optical-flow_amd", "csrc")
GOLDEN = os.path.join(ROOT, "tests", "golden", "isa_hashes.txt")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "k_lane8_*.hip")) + glob.glob(os.path.join(CSRC, "k_cols8_*.hip")))


def kernels_of(src, extra_flags=()):
    """[(kernel name, hash of its instructions, vgprs, vgpr spills, sgpr spills, scratch bytes)] of one source file."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "out.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{CSRC}", "--offload-arch=gfx950", "-fno-fast-math",
                        "-S", "--offload-device-only", *extra_flags, src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        s = open(out).read()
    meta = {}
    for blk in re.findall(r"- \.agpr_count.*?\.wavefront_size:\s+\d+", s, re.S):
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
        meta[g("name")] = (int(g("vgpr_count")), int(g("vgpr_spill_count")), int(g("sgpr_spill_count")), int(g("private_segment_fixed_size")))
    found = []
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M):
        # instructions only: no comment lines or trailing comments, and basic-block labels without the function's index in
        # the file (.LBB0_7 / .LBB1_7: a kernel added in front renumbers them)
        body = "\n".join(re.sub(r"BB\d+_", "BB_", l.split(";")[0].rstrip()) for l in m.group(2).split("\n") if not l.strip().startswith(";"))
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"aof::(\(anonymous namespace\)::)?", "", name)
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        found.append((name, hashlib.md5(body.encode()).hexdigest()[:12]) + meta.get(m.group(1), (-1, -1, -1, -1)))
    return found


def table(srcs=None):
    srcs = srcs or sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        per = list(ex.map(kernels_of, srcs))
    lines = []
    for src, ks in zip(srcs, per):
        for name, h, v, vs, ss, scratch in ks:
            lines.append(f"{os.path.basename(src):26s} {h}  vgpr {v:3d}  spills {vs}/{ss}  scratch {scratch} B  {name}")
    return lines


def main():
    files = [a for a in sys.argv[1:] if not a.startswith("--")]   # (explicit sources: any kernel file, e.g. k_coarse.hip)
    lines = table(files or None)
    if "--update" in sys.argv:
        with open(GOLDEN, "w") as f:
            f.write("# tools/isa_hashes.py --update: hash of the emitted gfx950 instructions of every lane-per-block 8x8 kernel (one per translation unit)\n")
            f.write("\n".join(lines) + "\n")
        print(f"wrote {GOLDEN}")
    elif "--check" in sys.argv:
        want = [l.rstrip("\n") for l in open(GOLDEN) if not l.startswith("#")]
        bad = [(a, b) for a, b in zip(sorted(want), sorted(lines)) if a != b]
        for a, b in bad:
            print(f"- {a}\n+ {b}")
        sys.exit(1 if bad or len(want) != len(lines) else 0)
    else:
        print("\n".join(lines))


if __name__ == "__main__":
    main()
