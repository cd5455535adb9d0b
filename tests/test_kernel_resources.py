"""The built library's kernels, as the compiler allocated them (tools/kernel_resources.py reads the code objects' metadata):
no kernel of the default path spills vector registers or uses scratch memory.  A spill does not fail a parity test -- it once
halved the 1280x720 TV-L1 throughput silently (the one-wave row pipeline, 19-21 spilled registers) -- so it is held here."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

# tested options that are known to spill a few registers and are not on any default path
ALLOWED_TO_SPILL = ("k_iter_stream<3, 5, 2,",)


@pytest.fixture(scope="module")
def kernels():
    import __graft_entry__ as entry
    entry.build()
    import kernel_resources
    rows = kernel_resources.kernels()
    assert len(rows) > 50, "the code objects of libva_hip.so were not found"
    return rows


def test_every_default_kernel_is_free_of_vector_spills_and_scratch(kernels):
    bad = [(r["name"], r["vgpr_spill"], r["scratch"]) for r in kernels
           if (r["vgpr_spill"] or r["scratch"]) and not r["name"].startswith(ALLOWED_TO_SPILL)]
    assert not bad, bad


def test_the_hot_kernels_keep_their_occupancy(kernels):
    by = {r["name"]: r for r in kernels}
    # two waves per SIMD for the row pipeline (<= 256 registers) in both forms; the two-group conv kernels too (512 threads)
    # (and the one-image kernel of the 14 x 14 layers: four computing + four loader waves = two per SIMD)
    for name, limit in (("k_iter_stream<2, 8, 2, false, 1>", 256), ("k_iter_stream<2, 10, 1, false, 1>", 256), ("k_conv3x3_pp_bf16<4, false>", 256),
                        ("k_conv3x3_pp_bf16<4, true>", 256)):
        assert name in by, (name, sorted(by)[:5])
        assert by[name]["vgpr"] + by[name]["agpr"] <= limit, by[name]
    # the four-wave row pipelines: three waves per SIMD (<= 168 registers), two workgroups per CU (<= 80 KB of LDS)
    # (4 x 4: <= 168 registers and <= 53.3 KB of LDS, i.e. three workgroups per CU; 4 x 5: its 60 KB allow two, <= 256 registers)
    r44, r45 = by["k_iter_stream<2, 4, 4, false, 1>"], by["k_iter_stream<2, 5, 4, false, 1>"]
    assert r44["vgpr"] + r44["agpr"] <= 168 and r44["lds"] <= 54613 and r44["max_wg"] == 256, r44
    assert r45["vgpr"] + r45["agpr"] <= 256 and r45["lds"] <= 81920 and r45["max_wg"] == 256, r45
    # 4 x 3 levels (the wide levels): four waves per SIMD, three workgroups per CU
    r43 = by["k_iter_stream<2, 3, 4, false, 1>"]
    assert r43["vgpr"] + r43["agpr"] <= 128 and r43["lds"] <= 54613, r43
    img14 = [r for r in kernels if "k_conv3x3_img14" in r["mangled"]]  # (llvm-cxxfilt does not demangle the bf16 instantiations)
    assert len(img14) == 4 and all(r["vgpr"] + r["agpr"] <= 256 and r["lds"] <= 163840 and r["max_wg"] == 512 for r in img14), img14
    # four workgroups per CU for the tap-major bf16 kernel (<= 128 registers, <= 40 KB of LDS)
    for name in ("k_conv3x3_mfma_bf16<2, false, false, 1>", "k_conv3x3_mfma_bf16<2, true, false, 1>"):
        assert by[name]["vgpr"] + by[name]["agpr"] <= 128 and by[name]["lds"] <= 40960, by[name]
    # the row pipeline: four two-wave jobs per CU (<= 40 KB of LDS each); the classifier's weight stream: >= 4 workgroups per CU
    assert by["k_iter_stream<2, 8, 2, false, 1>"]["lds"] <= 40960 and by["k_fc_splitk"]["vgpr"] + by["k_fc_splitk"]["agpr"] <= 128
    # LDS budgets: one 160 KB CU
    assert max(r["lds"] for r in kernels) <= 163840
