"""The lane-per-block 8x8 kernels are compiled ONE per translation unit (aof_lane8_kernels.hpp: the ISA of a kernel
depended on which other kernels of its translation unit shared its instantiation of search_block), and the instructions
each compiles to are pinned here: a change that moves a kernel's ISA fails this test until tests/golden/isa_hashes.txt
is regenerated (tools/isa_hashes.py --update) in the same commit -- so no kernel's code moves unnoticed.  CPU-side:
hipcc cross-compiles gfx950 without a GPU."""
import importlib.util
import os
import shutil
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_hashes", os.path.join(ROOT, "tools", "isa_hashes.py"))
isa = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa)

pytestmark = pytest.mark.skipif(not os.path.exists(isa.HIPCC), reason="hipcc not installed")


@pytest.fixture(scope="module")
def compiled():
    return isa.table()


def test_every_lane8_translation_unit_holds_exactly_one_kernel(compiled):
    per_file = {}
    for line in compiled:
        per_file.setdefault(line.split()[0], []).append(line)
    assert len(per_file) == len(isa.sources()) == 16
    for f, lines in per_file.items():
        assert len(lines) == 1, (f, lines)


def test_no_kernel_spills_and_all_keep_four_waves_per_simd(compiled):
    for line in compiled:
        w = line.split()
        vgpr_spills, sgpr_spills = (int(x) for x in w[5].split("/"))
        # vgprs; no vector register spilled, no scratch memory; a handful of scalar registers parked in VGPR lanes is harmless
        assert int(w[3]) <= 128 and vgpr_spills == 0 and sgpr_spills <= 8 and w[7] == "0", line


def test_isa_matches_the_committed_hashes(compiled):
    want = sorted(l.rstrip("\n") for l in open(isa.GOLDEN) if not l.startswith("#"))
    got = sorted(compiled)
    moved = [(a, b) for a, b in zip(want, got) if a != b]
    assert len(want) == len(got) and not moved, (
        "the emitted ISA of these kernels changed; if that is intended, run tools/isa_hashes.py --update and commit "
        "tests/golden/isa_hashes.txt with the change:\n" + "\n".join(f"- {a}\n+ {b}" for a, b in moved))


def test_another_kernel_in_the_translation_unit_does_not_move_the_headline_kernels(compiled):
    """What the isolation is for: a kernel added to a translation unit must not change its neighbour.  With one
    kernel per unit there is no neighbour left that shares an instantiation; an unrelated kernel added to the headline
    kernels' units leaves their instructions alone."""
    golden = {l.split()[0]: l.split()[1] for l in compiled}
    dummy = ("\nnamespace aof { __global__ void k_dummy_neighbour(uint32_t *out, const uint32_t *in, uint32_t n)\n"
             "{ const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = __builtin_amdgcn_sad_u8(in[i], in[n - 1 - i], i); } }\n")
    with tempfile.TemporaryDirectory() as tmp:
        for f in ("k_lane8_search_ff.hip", "k_cols8_search_f.hip"):
            dst = os.path.join(tmp, f)
            shutil.copy(os.path.join(isa.CSRC, f), dst)
            with open(dst, "a") as fh:
                fh.write(dummy)
            ks = {name: h for name, h, *_ in isa.kernels_of(dst)}
            assert len(ks) == 2 and "k_dummy_neighbour" in ks, ks
            mine = [h for name, h in ks.items() if name != "k_dummy_neighbour"]
            assert mine == [golden[f]], (f, ks, golden[f])
