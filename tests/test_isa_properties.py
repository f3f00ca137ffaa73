"""Compile-only checks of the gfx950 code the design relies on (hipcc cross-compiles without a GPU): the raster kernels
keep their per-Gaussian operands in SGPRs (scalar loads), the backward uses the permlane-swap transposed reduction,
nothing spills to scratch, and the sky backward keeps separate LDS / global atomics."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "street-gaussians-ns_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _asm(tmp_path_factory, name):
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    extra = ["-fno-slp-vectorize"] if name == "raster" else []      # as csrc/Makefile builds it
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", *extra,
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only", "-o", str(out),
           os.path.join(CSRC, name + ".hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return out.read_text()


def _kernels(asm):
    """kernel symbol -> its text (between the label and s_endpgm) and its resource footer."""
    res = {}
    for m in re.finditer(r"^(_Z\w+):.*?s_endpgm(.*?)(?=^_Z\w+:|\Z)", asm, re.S | re.M):
        res[m.group(1)] = m.group(0)
    return res


@pytest.fixture(scope="module")
def raster_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    return _asm(tmp_path_factory, "raster")


def test_raster_kernels_have_no_scratch_and_scalar_operands(raster_asm):
    ks = _kernels(raster_asm)
    fwd = [t for k, t in ks.items() if "raster_fwd_kernel" in k]
    bwd = [t for k, t in ks.items() if "raster_bwd_kernel" in k or "raster_bwd_short_kernel" in k]
    # (round 6: what ships is what runs — the "stream" mode, the forced one- / four-wave shapes and the MFMA reduction are
    # gone)  forward, tile sizes other than 16: exact x {3 channels, + depth channel}; backward: exact x reduce (2) x
    # {in-kernel adaptive split, long-walk half of the two-kernel scheme} + its short-walk half (own kernel)
    assert len(fwd) == 4 and len(bwd) == 12
    for t in fwd + bwd:
        assert re.search(r"ScratchSize: 0\b", t), "a raster kernel spills to scratch"
        assert "s_load_dwordx8" in t and "s_load_dwordx4" in t     # 48-byte row / record in SGPRs
    assert all("ds_read_b128" in t for t in fwd)                   # LDS-batched long-list path compiled in
    assert "v_mfma" not in raster_asm and "pack_records_kernel" not in raster_asm


def test_packed_forward_runs_on_the_packed_fp32_pipe(raster_asm):
    """The default forward (two pixels per lane) must really be packed: the quadratic form, alpha, the transmittance
    update and the three colour sums are v_pk_* instructions, and nothing spills."""
    ks = _kernels(raster_asm)
    pk = {k: t for k, t in ks.items() if "raster_fwd_pk_kernel" in k}
    assert len(pk) == 8                                            # exact x depth channel x group accumulations
    groups = {k: t for k, t in pk.items() if re.search(r"pk_kernelILb[01]ELb[01]ELb1EEE", k)}
    assert len(groups) == 4                                        # (r04) sgn_raster_fwd_groups: exact x depth channel
    for k, t in groups.items():
        # the walk with the two group accumulations AND the plain walk (tiles without an entry of the own-list group)
        # are both in the kernel; the ten-pointer descriptor of its first version spilled SGPRs to scratch memory
        assert re.search(r"ScratchSize: 0\b", t), "the group forward spills to scratch"
        plain = next(t2 for k2, t2 in pk.items() if k2 == k.replace("ELb1EEE", "ELb0EEE"))
        assert t.count("v_pk_fma_f32") > 2 * plain.count("v_pk_fma_f32") - 4, k
    pk = {k: t for k, t in pk.items() if k not in groups}
    for k, t in pk.items():
        assert re.search(r"ScratchSize: 0\b", t), "the packed forward spills to scratch"
        assert "s_load_dwordx8" in t and "ds_read_b128" in t      # scalar-chase and LDS-batched paths both compiled in
        # two code paths (scalar chase, LDS batches) x (3 colour + 2 quadratic-form) packed FMAs, + the 1-px long-tile body
        assert t.count("v_pk_fma_f32") >= 10 and t.count("v_pk_mul_f32") >= 12, k
    fast = next(t for k, t in pk.items() if "ILb0ELb0ELb0E" in k)
    assert fast.count("v_exp_f32") >= 4                            # hardware exp, two per entry and path
    # the depth channel (r03) is ONE more packed fma per entry and path in the two-pixel body, nothing else
    deep = next(t for k, t in pk.items() if "ILb0ELb1ELb0E" in k)
    assert 0 < deep.count("v_pk_fma_f32") - fast.count("v_pk_fma_f32") <= 3
    assert re.search(r"ScratchSize: 0\b", deep)


def test_short_walk_backward_is_held_at_four_waves_and_not_slp_packed(raster_asm):
    """raster.hip is built without the SLP vectoriser (a packed FP32 instruction costs two single ones on gfx950, the
    vectoriser's v_mov shuffles come on top), and the short-walk kernel must not grow past four waves per SIMD
    (it would starve the concurrently running long-walk kernel): the occupancy attribute must have taken."""
    ks = _kernels(raster_asm)
    short = {k: t for k, t in ks.items() if "raster_bwd_short_kernel" in k}
    assert len(short) == 4
    for k, t in short.items():
        assert "v_pk_fma_f32" not in t and "v_pk_mul_f32" not in t, k
        m = re.search(r"; Occupancy: (\d+)", t)
        assert m and int(m.group(1)) == 4, (k, m and m.group(1))


def test_backward_uses_the_permlane_swap_reduction(raster_asm):
    ks = _kernels(raster_asm)
    for k, t in ks.items():
        if "raster_bwd_kernel" not in k and "raster_bwd_short_kernel" not in k:
            continue
        reduce_mode = int(re.search(r"raster_bwd_(?:short_)?kernelILb[01]ELi([01])E", k).group(1))
        swaps = t.count("v_permlane32_swap") + t.count("v_permlane16_swap")
        if reduce_mode == 1:
            assert swaps == 16 and t.count("row_half_mirror") >= 6        # 8 swaps + 12 DPP adds per code path, 2 paths
        else:
            assert swaps == 0
        assert "v_mfma" not in t


def test_sky_backward_keeps_lds_and_global_atomics_apart(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    asm = _asm(tmp_path_factory, "cubemap")
    assert "ds_add_f32" in asm and "global_atomic_add_f32" in asm
    assert "flat_atomic_add_f32" not in asm     # the compiler once merged both arms into one flat atomic (1.3 ms)


def test_streaming_kernels_issue_their_loads_before_the_first_lds_write(tmp_path_factory):
    """A plain `for (...) { load; lds_store }` loop compiles to one HBM round trip per iteration (s_waitcnt vmcnt(0)
    after every global_load): the SH stage-in and the radix scatter request everything first."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    sh = _kernels(_asm(tmp_path_factory, "sh"))
    fwd = next(t for k, t in sh.items() if "sh_fwd_kernelILi16ELi4E" in k)
    fast = fwd[fwd.index("global_load_dwordx4"):]                 # the aligned fast path: 12 float4 per lane
    first_store = min(i for i in (fast.find("ds_write"), fast.find("ds_store")) if i >= 0)
    assert fast[:first_store].count("global_load_dwordx4") >= 10     # (the scheduler may sink one or two)
    # the grid-stride loop prefetches the NEXT span: a second batch of twelve float4 loads after the LDS writes
    assert fwd.count("global_load_dwordx4") >= 24 and "s_cbranch" in fwd
    rs = _kernels(_asm(tmp_path_factory, "radix_sort"))
    scat = next(t for k, t in rs.items() if "rs_scatter_kernelIjLb1ELi8ELi16ELb1ELb0E" in k)   # ATOMIC = true instantiation
    first_wait = scat.index("s_waitcnt vmcnt(0)")
    # 16 keys + 16 values + the digit total and the scanned-table column of the thread's digit: all in flight before the
    # ranking starts (the last two used to be requested after the ranking barrier)
    assert scat[:first_wait].count("global_load_dword") == 34
    # the per-wave counters are reached through an LDS pointer: as a generic volatile pointer they compiled to
    # flat_load / flat_store sc0 sc1 + s_waitcnt vmcnt(0), twice per key
    for k, t in rs.items():
        assert "flat_load" not in t and "flat_store" not in t, k
    # ranking: one returning LDS atomic per key (16 keys per thread in this instantiation), no ballot-match code
    assert scat.count("ds_add_rtn_u32") == 16 and "v_bitop3_b32" not in scat
    # ... and the documented ballot-match ranking is compiled next to it (what runs until sgn_sort_selftest has passed)
    ballot = next(t for k, t in rs.items() if "rs_scatter_kernelIjLb1ELi8ELi16ELb0ELb0E" in k)
    assert "ds_add_rtn_u32" not in ballot and "v_bitop3_b32" in ballot and "v_mbcnt_hi_u32_b32" in ballot
