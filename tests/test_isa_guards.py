"""Assembly-level guards for hand-written inline asm whose correctness depends on what the COMPILER puts around it.

nms.hip routes the lane select of v_writelane through M0 (suppression_row_pair): the compiler rejects "m0" in a clobber
list, so the code saves M0 before the 64 rows of a tile and restores it afterwards, and relies on nothing else between
the pair reading or writing M0.  This test compiles the TU for gfx950 exactly like the Makefile and
checks, in the ISA, that between every `s_mov_b32 sN, m0` (save) and the matching `s_mov_b32 m0, sN` (restore) the only
instructions that touch M0 are our own `s_mov_b32 m0, <imm>` and `v_writelane_b32 ..., m0` — and that no instruction with
an IMPLICIT M0 operand (LDS-DMA, movrel, GWS, sendmsg, interp) appears there.  A compiler upgrade that starts using M0 in
that region fails here instead of silently corrupting suppression masks."""
import os
import re
import subprocess

import pytest

from helpers import ROOT

CSRC = os.path.join(ROOT, "vision_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
IMPLICIT_M0 = re.compile(r"^\s*(s_movrel|v_movrel|ds_gws|s_sendmsg|v_interp|buffer_load\S*\s.*\blds\b|global_load_lds|s_load_lds)")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_nms_m0_save_restore_regions_are_clean(tmp_path):
    out = tmp_path / "nms.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           f"-I{os.path.join(ROOT, 'include')}", "-S", "--cuda-device-only", os.path.join(CSRC, "nms.hip"), "-o", str(out)]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    lines = out.read_text().splitlines()
    save_re = re.compile(r"^\s*s_mov_b32\s+(s\d+),\s*m0\b")
    regions = writes = 0
    i = 0
    while i < len(lines):
        m = save_re.match(lines[i])
        if not m:
            i += 1
            continue
        sreg = m.group(1)
        restore_re = re.compile(rf"^\s*s_mov_b32\s+m0,\s*{sreg}\b")
        j = i + 1
        while j < len(lines) and not restore_re.match(lines[j]):
            ln = lines[j].split(";")[0]
            assert not re.match(r"^\s*s_endpgm", ln), f"M0 saved at line {i + 1} but never restored before s_endpgm"
            assert not IMPLICIT_M0.match(ln), f"line {j + 1}: instruction with an implicit M0 operand inside a save/restore region: {ln.strip()}"
            if re.search(r"\bm0\b", ln):
                ok = re.match(r"^\s*s_mov_b32\s+m0,\s*(0x[0-9a-f]+|\d+)\s*$", ln) or re.match(r"^\s*v_writelane_b32\s+v\d+,\s*(s\d+|vcc_lo|vcc_hi),\s*m0\s*$", ln)
                assert ok, f"line {j + 1}: unexpected use of M0 inside a save/restore region: {ln.strip()}"
                writes += 1
            # the save register must stay untouched until the restore
            assert not re.match(rf"^\s*\S+\s+{sreg}\b(?!\s*,\s*m0)", ln) or "s_mov_b32 m0" in ln, f"line {j + 1}: save register {sreg} overwritten: {ln.strip()}"
            j += 1
        assert j < len(lines), f"M0 saved at line {i + 1} but never restored"
        regions += 1
        i = j + 1
    # the fast tile path exists in every tile kernel (mask tiles, segmented, small segments, diag / keyed variants)
    assert regions >= 3 and writes >= 3 * 96, (regions, writes)


def _kernel_resources(src, tmp_path):
    """Compile one TU like the Makefile and return {kernel name: {vgpr, spill, lds, scratch}} from its code-object metadata."""
    out = tmp_path / (os.path.basename(src) + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           f"-I{os.path.join(ROOT, 'include')}", "-S", "--cuda-device-only", src, "-o", str(out)]
    if src.endswith("deform_conv2d.hip"):
        cmd.insert(1, "-fno-slp-vectorize")   # as in the Makefile
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    text = out.read_text()
    res = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:", text, re.S):
        blk = m.group(0)
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)

        def num(key, blk=blk):
            mm = re.search(rf"\.{key}:\s+(\d+)", blk)
            return int(mm.group(1)) if mm else 0
        res[name] = dict(vgpr=num("vgpr_count"), spill=num("vgpr_spill_count"), lds=num("group_segment_fixed_size"),
                         scratch=num("private_segment_fixed_size"))
    return text, res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_occupancy_assumptions_of_the_hot_kernels(tmp_path):
    """DESIGN.md quotes occupancies that follow from compiler output: the fp32 RoIAlign DMA kernel must keep FOUR 256-thread
    workgroups per CU (LDS <= 40 KB per workgroup, <= 128 VGPRs) after its output staging block was added, the 16-bit
    channels-last deform_conv2d kernel two 512-thread workgroups (<= 128 VGPRs, < 64 KB of LDS), none of them may spill, and
    the three-instruction row groups of the DMA kernel need their own counted wait (vmcnt(3), never the vmcnt(4) of the other
    instantiations)."""
    text, roi = _kernel_resources(os.path.join(CSRC, "roi_align.hip"), tmp_path)
    dma = {k: v for k, v in roi.items() if "roi_align_fwd_ms_dmaIfLi7ELi7ELi2E" in k}
    assert len(dma) == 1, list(roi)
    for r in dma.values():
        assert r["spill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 96 and 0 < r["lds"] <= 40 * 1024, r
        assert 4 * r["lds"] <= 160 * 1024
    # round 6: the default 7 x 7 launch carries the wave path for declined units inline — the union of both must keep four
    # waves per SIMD (<= 128 VGPRs) and four workgroups per CU (4 x LDS <= 160 KB)
    inl = {k: v for k, v in roi.items() if "roi_align_fwd_ms_dma_inlIfLi7ELi7ELi2E" in k}
    assert len(inl) == 1, list(roi)
    for r in inl.values():
        assert r["spill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 128 and 4 * r["lds"] <= 160 * 1024, r
    # ... and the launch that also carries the detector step's NMS workgroups (the LDS of the two jobs is one union, the NMS body
    # must not bring static LDS of its own — __syncthreads_and / _or did: 256 bytes, three workgroups per CU instead of four)
    step = {k: v for k, v in roi.items() if "roi_align_fwd_ms_dma_inl_stepI" in k}
    assert len(step) == 3, list(roi)
    for r in step.values():
        assert r["spill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 128 and 4 * r["lds"] <= 160 * 1024, r
    big = {k: v for k, v in roi.items() if "roi_align_fwd_ms_dmaIfLi14ELi14ELi2E" in k}
    assert len(big) == 1 and all(v["vgpr"] <= 128 and v["spill"] == 0 and v["scratch"] == 0 for v in big.values()), big
    assert re.search(r"s_waitcnt vmcnt\(3\)", text) and re.search(r"s_waitcnt vmcnt\(4\)", text)
    _, dcn = _kernel_resources(os.path.join(CSRC, "deform_conv2d.hip"), tmp_path)
    cl = {k: v for k, v in dcn.items() if "dcn_fwd_mfma_16_cl" in k and "Li4ELi2ELi2ELi1E" in k}
    assert len(cl) == 2, list(dcn)          # fp16 and bf16
    for r in cl.values():
        assert r["spill"] == 0 and r["vgpr"] <= 128, r
    dw = {k: v for k, v in dcn.items() if "dcn_fwd_depthwise3x3IfLi75E" in k}
    assert len(dw) == 1 and all(v["spill"] == 0 and v["vgpr"] <= 128 for v in dw.values()), dw


def _kernel_bodies(text, needle):
    """{mangled name: ISA text up to s_endpgm} of the kernels whose name contains `needle`."""
    out = {}
    for m in re.finditer(r"^(_ZN4tvmi\S+):\s+; @\S+\n(.*?)s_endpgm", text, re.S | re.M):
        if needle in m.group(1):
            out[m.group(1)] = m.group(2)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_roi_align_dma_loop_keeps_its_prefetch_in_flight(tmp_path):
    """The LDS-DMA forward kernels double-buffer by hand: the DMAs of pass p+1 are issued, a COUNTED `s_waitcnt vmcnt(N)` retires
    pass p, and the arithmetic of pass p runs while p+1 is in flight.  Two things the compiler did to that loop in round 3,
    both of which put `s_waitcnt vmcnt(0)` in front of every channel's arithmetic (= wait for the prefetch just issued):
      * {l, h} of axis_sample_shifted lived in a scratch pair (conditional stores through references) and the scratch loads'
        VM-counter dependency was carried into the loop;
      * plain C++ stores into the LDS output staging block are 'may alias a pending LDS-DMA' for the waitcnt pass.
    Guard: no scratch, and the only vmcnt(0) waits are the ones written in the source (one per row-group instantiation: the
    last pass / the single-buffer NRG=8 form, plus the RoI loads of the 16-bit single-level entry).  Also the tap pairing:
    8 pair reads per channel and instantiation, no split halves."""
    src = os.path.join(CSRC, "roi_align.hip")
    text, _ = _kernel_resources(src, tmp_path)
    bodies = _kernel_bodies(text, "roi_align_fwd_")
    dma = {k: v for k, v in bodies.items() if "_dmaI" in k}
    assert len(dma) == 12, sorted(dma)      # {single level, multi-scale} x {fp32, fp16, bf16} x {7x7, 14x14}
    for name, body in dma.items():
        assert "scratch_" not in body, name
        n0 = len(re.findall(r"s_waitcnt vmcnt\(0\)", body))
        assert n0 <= 7, (name, n0)      # 5 written ones + the RoI loads of the 16-bit single-level entry + the worklist atomic
        assert not re.search(r"s_waitcnt vmcnt\(0\) lgkmcnt", body), name     # a compiler-made combined wait inside the channel loop
        if "IfLi7ELi7E" in name:     # row-group forms 1, 2, 3 x 2 buffers, 4 x 2 buffers, 8: 7 loop bodies of 8 pair reads, no address VALU
            assert len(re.findall(r"ds_read2_b32", body)) == 7 * 8, name
        if "IfLi14ELi14E" in name:
            assert len(re.findall(r"ds_read2_b32", body)) == 5 * 8 * 4, name


def _synchronous_loads(body, window=6):
    """Global loads that are followed by `s_waitcnt vmcnt(0)` within `window` instructions: a load made synchronous (tools/isa_waits.py)."""
    n, left = 0, 0
    for ln in body.splitlines():
        t = ln.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("global_load") or op.startswith("flat_load"):
            left = window
        elif op == "s_waitcnt" and "vmcnt(0)" in t and left > 0:
            n, left = n + 1, 0
        elif left > 0:
            left -= 1
    return n


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_roi_align_wave_kernels_stage_their_window_asynchronously(tmp_path):
    """The register-staged wave kernels (generic pooled shapes / adaptive sampling, the mop-up launch, calls without a workspace)
    load a window with 12-16 loads in flight per lane.  Rounds 1-4 guarded every load with the wave-uniform `if (rg < nrg)`: each
    became a branch + load + `s_waitcnt vmcnt(0)` (7 of 16 synchronous, profiles/r04_isa_waits.txt).  Guard: at most two loads in
    a kernel are waited for at once (the RoI row), the loads use the SGPR-base + 32-bit-offset form (not a 64-bit VGPR pair per
    row group), and the 7x7 / generic kernels stay within 168 VGPRs (three waves per SIMD; they need 121-129, their 10 KB of LDS per
    wave allows four)."""
    text, res = _kernel_resources(os.path.join(CSRC, "roi_align.hip"), tmp_path)
    bodies = {k: v for k, v in _kernel_bodies(text, "roi_align_fwd_").items() if "_waveI" in k}
    assert len(bodies) == 18, sorted(bodies)     # {single level, multi-scale} x {fp32, fp16, bf16} x {7x7, 14x14, generic}
    for name, body in bodies.items():
        assert "scratch_" not in body, name
        assert _synchronous_loads(body) <= 2, (name, _synchronous_loads(body))
        staged = re.findall(r"global_load_(?:ushort|dword) v\d+, v\d+, s\[\d+:\d+\]", body)
        assert len(staged) >= 100, (name, len(staged))
        if "Li14ELi14E" not in name:
            assert res[name]["vgpr"] <= 168 and res[name]["spill"] == 0, (name, res[name])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_nms_chain_of_the_step_fits_next_to_the_roi_align_kernel(tmp_path):
    """The two halves of the detection step overlap on two streams only because EVERY launch of the NMS / packing chain can start
    on a CU that already holds four workgroups of the RoIAlign forward: 160 KB - 4 x its LDS is what is left (4 KB today), 16 of
    the 32 wave slots are free.  Round 4's chain had a 21 KB and a 32 KB launch in it: they waited 100-195 us in the dispatcher and
    the chain finished BEHIND the RoIAlign launch (HISTORY.md 6.0).  Guard: the rank-counting score sort, the segment collect, the
    four-tile kernel, the sweep and the packing kernel each need at most that much LDS and at most 16 waves."""
    _, roi = _kernel_resources(os.path.join(CSRC, "roi_align.hip"), tmp_path)
    dma = [v for k, v in roi.items() if "roi_align_fwd_ms_dmaIfLi7ELi7ELi2E" in k]
    assert len(dma) == 1
    left = 160 * 1024 - 4 * dma[0]["lds"]
    assert left >= 4096, left
    _, nms = _kernel_resources(os.path.join(CSRC, "nms.hip"), tmp_path)
    _, post = _kernel_resources(os.path.join(CSRC, "postprocess.hip"), tmp_path)
    want = {"sort_scores_desc_rank": nms, "nms_small_seg_collectIf": nms, "nms_small_seg_tiles4If": nms, "nms_small_seg_sweep": nms,
            "pack_detections_kernel": post}
    for frag, table in want.items():
        hits = {k: v for k, v in table.items() if frag in k}
        assert len(hits) == 1, (frag, list(hits))
        for k, v in hits.items():
            assert v["lds"] <= left and v["spill"] == 0 and v["scratch"] == 0 and v["vgpr"] <= 64, (k, v, left)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_nms_step_kernel_shape(tmp_path):
    """The one-launch step kernel (nms_step_fused, round 6) runs ONCE per call on an otherwise idle slice of the chip: every phase
    executes cold code, so its size is a cost (a fully unrolled first version was 22,000 instructions = 130 KB and no faster than
    the five launches it replaced).  Guard: no spills / scratch, at most 128 VGPRs, under 32 KB of LDS, under 13,000 instructions,
    and the counting loops compare score words with saturating subtracts (no v_cmp -> v_addc condition-code chain per key)."""
    text, nms = _kernel_resources(os.path.join(CSRC, "nms.hip"), tmp_path)
    hits = {k: v for k, v in nms.items() if "nms_step_fused" in k}
    assert len(hits) == 1, list(nms)
    (name, r), = hits.items()
    assert r["spill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 128 and r["lds"] <= 32 * 1024, r
    body = list(_kernel_bodies(text, "nms_step_fused").values())
    assert len(body) == 1
    insts = [ln for ln in body[0].splitlines() if ln.startswith("\t") and not ln.startswith("\t.") and not ln.strip().startswith(";")]
    assert len(insts) < 13000, len(insts)
    assert sum("v_sub_u32" in ln and "clamp" in ln for ln in insts) >= 8, "saturating-subtract counting loops not found"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_resize_backward_blocks_are_loaded_together(tmp_path):
    """The gather backward of the resize modes (resize.hip, upsample2d_bwd_vec_kernel<T, R, CC, P>) beat ATen's scatter kernels on
    up-scales only once the R x P row loads of a block were issued together as CC-element vector loads (the first version's
    loads sat under per-lane range predicates, four in flight: FPN nearest 2x 0.172 ms against ATen's 0.095, now 0.081).
    Guard: every instantiation loads its block rows as ONE vector load each (fp32: dwordx2 / dwordx4; 16-bit: dword / dwordx2 at
    2-byte alignment), at least R * P of them in a loop body, nothing in scratch, within 96 VGPRs."""
    text, res = _kernel_resources(os.path.join(CSRC, "resize.hip"), tmp_path)
    bodies = _kernel_bodies(text, "upsample2d_bwd_vec_kernel")
    assert len(bodies) == 12, sorted(bodies)      # {fp32, fp16, bf16} x {(2,2,8), (4,2,4), (2,4,4), (4,4,4)}
    for name, body in bodies.items():
        m = re.search(r"Li(\d)ELi(\d)ELi(\d)E", name)
        R, CC, P = (int(v) for v in m.groups())
        wide = {("f", 2): "dwordx2", ("f", 4): "dwordx4", ("h", 2): "dword", ("h", 4): "dwordx2"}[("f" if "kernelIf" in name else "h", CC)]
        n = len(re.findall(rf"global_load_{wide} ", body))
        assert n >= R * P, (name, wide, n)
        assert "scratch_" not in body and res[name]["spill"] == 0 and res[name]["vgpr"] <= 96, (name, res[name])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_roi_align_backward_owner_keeps_its_prefetch_in_flight(tmp_path):
    """Round 4: the tile-owner backward lost 20-30 % to loads the compiler had made synchronous — the grads prefetch of entry
    e + 1 was `v = 0; if (inside) v = load; else if (straddles) {...}` (a phi: its copy and an s_waitcnt vmcnt(0) sat right
    behind the load), it was issued in front of the AxD loads and behind an `if (e + 1 < total)`.  Guard: in the fp32 owner
    kernels no 16-byte load is followed by vmcnt(0) within a few instructions, and the FMAs of an entry start behind a COUNTED
    wait that leaves the prefetch in flight (7 AxD pairs + 2 prefetch pieces -> vmcnt(8) at 7x7; 14 + 7 -> vmcnt(20) at 14x14)."""
    text, res = _kernel_resources(os.path.join(CSRC, "roi_align_bwd.hip"), tmp_path)
    bodies = {k: v for k, v in _kernel_bodies(text, "roi_align_bwd_ownerI").items() if "_bigI" not in k}
    assert len(bodies) == 6, sorted(bodies)      # {fp32, fp16, bf16} x {7x7, 14x14}
    for name, body in bodies.items():
        assert "scratch_" not in body, name
        assert not re.search(r"global_load_dwordx4[^\n]*\n(?:[^\n]*\n){0,3}?\s*s_waitcnt vmcnt\(0\)", body), name
    b7 = next(v for k, v in bodies.items() if "IfLi7ELi7E" in k)
    b14 = next(v for k, v in bodies.items() if "IfLi14ELi14E" in k)
    assert re.search(r"s_waitcnt vmcnt\(8\)", b7) and re.search(r"s_waitcnt vmcnt\(20\)", b14)
    own7 = next(v for k, v in res.items() if "roi_align_bwd_ownerIfLi7ELi7E" in k)
    assert own7["vgpr"] <= 96 and own7["spill"] == 0, own7       # five 256-thread workgroups per CU


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_deform_conv2d_backward_slab_loops_are_asynchronous(tmp_path):
    """The fused backward: the owner data kernel must not spill (with all nine taps resident the allocator spilled
    the slab PREFETCH registers — a vmcnt(0) and a scratch store behind every load), and neither it nor the channels-last
    weight-gradient kernel may wait for a load right behind it (the phi copy of a conditional load / of `if (use_mask) m =
    mask[...]` did exactly that in front of the MFMAs)."""
    text, res = _kernel_resources(os.path.join(CSRC, "deform_conv2d_bwd.hip"), tmp_path)
    own = {k: v for k, v in res.items() if "dcn_bwd_data_ownI" in k}
    assert len(own) == 3, sorted(res)
    for k, r in own.items():
        assert r["spill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 256, (k, r)
    bodies = _kernel_bodies(text, "dcn_bwd_")
    for name, body in bodies.items():
        if "dcn_bwd_data_ownI" in name or ("dcn_bwd_weight_mfmaI" in name and "Lb1E" in name):
            assert "scratch_" not in body, name
            assert _synchronous_loads(body) == 0, (name, _synchronous_loads(body))
            # fp32 tensors: 32x32x2 fp32 steps; 16-bit tensors (round 5): ONE 32x32x16 step per accumulator block and K slab
            if "I6__half" in name:
                assert len(re.findall(r"v_mfma_f32_32x32x16_f16", body)) >= 4 and "v_mfma_f32_32x32x2_f32" not in body, name
            elif "I14__hip_bfloat16" in name:
                assert len(re.findall(r"v_mfma_f32_32x32x16_bf16", body)) >= 4 and "v_mfma_f32_32x32x2_f32" not in body, name
            else:
                assert len(re.findall(r"v_mfma_f32_32x32x2_f32", body)) >= 32, name


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_depthwise_deform_conv2d_packed_kernel_shape(tmp_path):
    """The packed depthwise forward (round 5): two 512-thread workgroups per CU need <= 128 VGPRs without a spill (a
    spill in the tap loop is a scratch load + s_waitcnt vmcnt(0) in front of the LDS reads — that also waits for the window
    prefetch); per chunk of four channels it must issue 36 16-byte corner reads, 72 packed ops (v_pk_mul / v_pk_fma with the
    op_sel broadcast of the corner weight) and take its weights from v_readlane, never from a scalar load inside the loop
    (scalar loads share lgkmcnt with the LDS reads and return out of order)."""
    text, res = _kernel_resources(os.path.join(CSRC, "deform_conv2d.hip"), tmp_path)
    fast = {k: v for k, v in res.items() if "dcn_fwd_depthwise3x3_pkI" in k and "Li80ELi3ELi3E" in k}
    assert len(fast) == 3, sorted(res)           # fp32, fp16, bf16
    for k, r in {k: v for k, v in res.items() if "dcn_fwd_depthwise3x3_pkI" in k}.items():
        assert r["spill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 128, (k, r)
    body = next(v for k, v in _kernel_bodies(text, "dcn_fwd_depthwise3x3_pkIfLi80ELi3ELi3E").items())
    assert len(re.findall(r"ds_read_b128", body)) == 36, len(re.findall(r"ds_read_b128", body))
    assert len(re.findall(r"v_pk_fma_f32", body)) == 54 and len(re.findall(r"v_pk_mul_f32", body)) == 18
    assert len(re.findall(r"v_readlane_b32", body)) >= 40
    # 12 corner reads in flight before the first blend of a tap group: a counted wait that leaves 11 of them outstanding
    assert re.search(r"s_waitcnt lgkmcnt\(11\)", body), "the tap groups' reads are no longer issued together"
    loop = body[body.index("ds_read_b128"):body.rindex("ds_read_b128")]
    assert "s_load_" not in loop and "scratch_" not in loop
