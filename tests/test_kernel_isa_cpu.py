"""ISA-level audit of the decode-step "dec" GEMM (csrc/swx_decstep.hip), run on the cross-compiled gfx950 assembly (no GPU).

The kernel hides its weight loads from hipcc in inline asm and counts their completion by hand (so that the MFMA loop can
start while the weight stream is still landing; with an LDS-DMA in flight hipcc would otherwise wait vmcnt(0) at the first
use of any load result).  What hipcc does NOT do for an asm statement (cdna_hip_programming.md 5.7) is checked here after
every build instead of by eye:
  1. no compiler-generated instruction reads or writes a register that an asm load targets before that register's own
     counted wait (a v_mov / spill of a not-yet-landed register would silently compute on garbage);
  2. between the first barrier (activation tile landed) and the last MFMA there is no `s_waitcnt vmcnt(0)`: the weight stream
     is consumed fragment by fragment, not drained;
  3. no scratch (spill) traffic in any instantiation.
"""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


@pytest.fixture(scope="module")
def asm_text():
    return _device_asm("swx_decstep.hip")


def _kernels(text):
    out, cur, name = {}, None, None
    for ln in text.split("\n"):
        m = re.match(r"^(_ZN\S*gemm_dec_f16\S*):", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(ln)
            if "s_endpgm" in ln:
                out[name] = cur
                cur = None
    return out


def test_dec_gemm_asm_loads_are_not_touched_before_their_wait(asm_text):
    kernels = _kernels(asm_text)
    assert len(kernels) >= 60, len(kernels)          # 6 depths x 3 row-tile counts x 5 epilogues, minus nothing
    for name, lines in kernels.items():
        pending, in_asm, nload, first_barrier, last_mfma, drains = {}, False, 0, None, None, []
        pf_pending, n_pf = set(), 0            # L2-prefetch loads (global_load_dword): results unused, registers held to the final wait
        red_pending, n_red = set(), 0          # DEC_TICKET: coherent slab reads of the in-launch reduction
        for i, ln in enumerate(lines):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if t.startswith("s_barrier") and first_barrier is None:
                first_barrier = i
            if t.startswith("v_mfma"):
                last_mfma = i
            if "scratch_" in t:
                pytest.fail(f"{name}: scratch traffic: {t}")
            if in_asm and t.startswith("global_load_dwordx4") and " sc1" in t:
                # DEC_TICKET (round 5): the last arriver's coherent reads of the K slices' slabs, issued after the weight stream is
                # long consumed; ONE asm `s_waitcnt vmcnt(0)` behind each batch of four releases them
                red_pending |= _regs(t.split()[1].rstrip(","))
                n_red += 1
                continue
            if in_asm and t.startswith("global_load_dwordx4"):
                for r in _regs(t.split()[1].rstrip(",")):
                    pending[r] = nload
                nload += 1
                continue
            if in_asm and t.startswith("global_load_dword "):
                pf_pending |= _regs(t.split()[1].rstrip(","))
                n_pf += 1
                continue
            if in_asm and t.startswith("s_waitcnt vmcnt(0)") and not pending:
                pf_pending.clear()               # the wave's last wait: every load has landed
                red_pending.clear()
                continue
            if not in_asm and red_pending and t and not t.startswith((";", ".")):
                used = set()
                for tk in re.findall(r"v\[\d+:\d+\]|v\d+", t):
                    used |= _regs(tk)
                hit = used & red_pending
                assert not hit, f"{name}: line {i}: `{t}` touches a slab read's register {sorted(hit)[:4]} before its wait"
            if not in_asm and pf_pending and t and not t.startswith((";", ".")):
                used = set()
                for tk in re.findall(r"v\[\d+:\d+\]|v\d+", t):
                    used |= _regs(tk)
                hit = used & pf_pending
                assert not hit, f"{name}: line {i}: `{t}` touches a prefetch load's register {sorted(hit)[:4]} while the load may be in flight"
            if in_asm and t.startswith("s_waitcnt vmcnt") and pending:
                oldest = min(pending.values())       # the waits are issued in load order, one fragment per wait
                for r in [r for r, k in pending.items() if k == oldest]:
                    del pending[r]
                continue
            if not in_asm and t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                drains.append(i)
            if not in_asm and pending and t and not t.startswith((";", ".")):
                used = set()
                for tk in re.findall(r"v\[\d+:\d+\]|v\d+", t):
                    used |= _regs(tk)
                hit = used & set(pending)
                assert not hit, f"{name}: line {i}: `{t}` touches asm-loaded registers {sorted(hit)[:4]} before their wait"
        assert nload in (12, 16, 20, 24, 32, 40), (name, nload)
        assert n_pf == 4, (name, n_pf)
        assert not pending and not pf_pending and not red_pending, (name, len(pending), len(pf_pending), len(red_pending))
        assert n_red in (0, 4, 8, 12), (name, n_red)                 # four slab reads per row tile of a DEC_TICKET instantiation
        if re.search(r"gemm_dec_f16ILi\d+ELi\d+ELi\d+ELi1EEE", name):
            # WPB = 1 (round 6): a workgroup of ONE wave -- hipcc drops the barrier instructions; the tile is waited for by the wave itself
            assert last_mfma is not None, name
        else:
            assert first_barrier is not None and last_mfma is not None and first_barrier < last_mfma, name
        early = [d for d in drains if d < last_mfma]
        assert not early, f"{name}: s_waitcnt vmcnt(0) before the last MFMA (lines {early[:3]}): the weight stream is drained"


_ASM_CACHE = {}


def _device_asm(src_name):
    """gfx950 assembly of one csrc file (device only), compiled once per test session"""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not present")
    if src_name not in _ASM_CACHE:
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(ROOT, "stable_ts_amd", "csrc", src_name)
            out = os.path.join(td, "k.s")
            subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
                                   "--cuda-device-only", src, "-o", out], cwd=td, stderr=subprocess.DEVNULL)
            _ASM_CACHE[src_name] = open(out).read()
    return _ASM_CACHE[src_name]


def _kernel_meta(src_name):
    """(kernel name -> dict of the metadata hipcc emits) for one csrc file, compiled for gfx950 (device only)"""
    text = _device_asm(src_name)
    meta = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size", text, re.S):
        blk = m.group(0)
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
                      for k in ("vgpr_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    return meta


@pytest.mark.parametrize("src,kernels,max_vgpr", [
    # the default tiled GEMMs: three workgroups per CU need <= 168 VGPRs and <= 53 KB of LDS each
    ("swx_gemm.hip", ["gemm_f16_glds_128", "gemm_f16_glds_64"], 168),
    # DTW generation 3 (x chunks in registers) and the register-resident logit filters (1024 threads: <= 128 VGPRs)
    ("swx_dtw.hip", ["swx_dtw4_kernelILi1ELb1E", "swx_dtw4_kernelILi2ELb0E"], 256),
    ("swx_decode.hip", ["decode_select_reg_kernel"], 128),
])
def test_hot_kernels_do_not_spill(src, kernels, max_vgpr):
    meta = _kernel_meta(src)
    for want in kernels:
        hits = {n: v for n, v in meta.items() if want in n}
        assert hits, (want, sorted(meta)[:5])
        for n, v in hits.items():
            assert v["private_segment_fixed_size"] == 0, (n, v)            # no scratch: nothing spilled
            assert v["vgpr_count"] <= max_vgpr, (n, v)
            if "gemm_f16_glds" in n:
                assert v["group_segment_fixed_size"] <= 53 * 1024, (n, v)


def test_ring_gemm_k_loop_waits_are_the_counted_ones():
    """gemm_f16_ring (csrc/swx_gemm.hip): the LDS-DMA of its operand ring is issued from inline asm and retired by hand --
    `s_waitcnt vmcnt((NST - 2) * L) lgkmcnt(0)` + `s_barrier` per K step, L = DMA instructions per wave and stage.  Audited
    on the compiled ISA of both instantiations (64- and 128-column tiles, depth 3): (1) between the first DMA and the last MFMA every `vmcnt` wait is one of
    the asm statements (an `s_waitcnt vmcnt` hipcc adds on its own there would drain the ring); (2) their immediates are
    exactly {(NST - 2) L, ..., L, 0}; (3) every asm wait is followed by the barrier; (4) no scratch, and the K loop holds
    2 x 4 x NJ MFMAs per step."""
    text = _device_asm("swx_gemm.hip")
    seen = 0
    for bnt, nst in ((64, 3), (128, 3)):
        m = re.search(rf"^(_ZN\S*gemm_f16_ringILi{bnt}ELi{nst}E[^\s:]*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
        assert m, (bnt, nst)
        seen += 1
        lines = m.group(2).split("\n")
        L = 4 + bnt // 32
        dma = [i for i, ln in enumerate(lines) if "global_load_lds_dwordx4" in ln]
        mfma = [i for i, ln in enumerate(lines) if "v_mfma_f32_16x16x32_f16" in ln]
        assert len(dma) == (nst - 1) * L + L, (bnt, nst, len(dma))          # prologue stages + the one refill in the loop
        assert len(mfma) == 2 * 4 * (bnt // 32), (bnt, nst, len(mfma))
        in_asm, waits = False, []
        for i in range(dma[0], max(mfma[-1], dma[-1])):      # the K loop's blocks (hipcc places the MFMA block first)
            ln = lines[i]
            if "#ASMSTART" in ln:
                in_asm = True
            elif "#ASMEND" in ln:
                in_asm = False
            w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", ln)
            if w:
                assert in_asm, f"ring<{bnt},{nst}>: hipcc's own `{ln.strip()}` inside the K loop"
                assert "lgkmcnt(0)" in ln and "s_barrier" in lines[i + 1], (bnt, nst, ln)
                waits.append(int(w.group(1)))
        assert sorted(set(waits)) == sorted({k * L for k in range(nst - 1)}), (bnt, nst, waits)
    assert seen == 2
    meta = _kernel_meta("swx_gemm.hip")
    ring = {n: v for n, v in meta.items() if "gemm_f16_ring" in n}
    assert len(ring) == 2
    for n, v in ring.items():
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_count"] <= 256, (n, v)


def test_big8_gemm_phases():
    """gemm_f16_big8 (the 256 x 256 tile kernel: a ring of eight half-tile slots, csrc/swx_gemm.hip): the compiled ISA must hold the schedule
    the ordering argument in the source is made for.  Three copies of a K tile (steady, second-last, last), each four phases of
    [fragment reads -> counted vmcnt -> barrier -> lgkmcnt(0) -> DMA issue -> 16 MFMAs -> barrier] with 12 / 4 / 8 / 0 reads; every
    `vmcnt` wait is one of the hand-written ones (10 after the prologue's 14 DMA instructions; 8, 8, -, 8 in the steady tile;
    8, 8, -, 4 and 2, 0, -, - in the last two); the DMA instructions of a phase come AFTER its first barrier (a slot is rewritten
    only once every wave has retired its reads of it) and its fragment reads BEFORE it; no scratch, <= 256 VGPRs."""
    text = _device_asm("swx_gemm.hip")
    m = re.search(r"^(_ZN\S*gemm_f16_big8E[^\s:]*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
    assert m
    lines = [ln.strip() for ln in m.group(2).split("\n")]
    ev, in_asm = [], False                     # the K loop as a string of events
    for ln in lines:
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", ln)
        if w:
            ev.append(("vm", int(w.group(1)), in_asm))
        elif ln.startswith("s_barrier"):
            ev.append(("bar", 0, in_asm))
        elif ln.startswith("global_load_lds_dwordx4"):
            ev.append(("dma", 0, in_asm))
        elif ln.startswith("ds_read_b128"):
            ev.append(("rd", 0, in_asm))
        elif ln.startswith("v_mfma_f32_16x16x32_f16"):
            ev.append(("mfma", 0, in_asm))
        elif ln.startswith(("global_load_", "global_store_", "ds_write")) :
            ev.append(("epi", 0, in_asm))
    first_epi = next(i for i, e in enumerate(ev) if e[0] == "epi")
    loop = ev[:first_epi]
    assert all(e[2] for e in loop if e[0] in ("vm", "bar", "dma")), "a wait / barrier / DMA of hipcc's own inside the K loop"
    # run-length encode
    rle = []
    for k, v, _ in loop:
        if k == "vm":
            rle.append(("vm", v))
        elif rle and rle[-1][0] == k and k != "bar":
            rle[-1] = (k, rle[-1][1] + 1)
        else:
            rle.append((k, 1))
    def phase(reads, wait, dma):
        out = []
        if reads:
            out.append(("rd", reads))
        if wait is not None:
            out.append(("vm", wait))
        out.append(("bar", 1))
        if dma:
            out.append(("dma", 2))
        out += [("mfma", 16), ("bar", 1)]
        return out
    def tile(waits, dmas):
        return phase(12, waits[0], dmas[0]) + phase(4, waits[1], dmas[1]) + phase(8, None, dmas[2]) + phase(0, waits[2], dmas[3])
    want = [("dma", 14), ("vm", 10), ("bar", 1), ("bar", 1)]                     # prologue, barrier, row group 1's extra barrier
    want += tile((8, 8, 8), (1, 1, 1, 1)) + tile((8, 8, 4), (1, 0, 0, 0)) + tile((2, 0, None), (0, 0, 0, 0))
    want += [("bar", 1)]                                                         # row group 0's extra barrier: the ring is the epilogue's now
    assert rle == want, [(a, b) for a, b in zip(rle, want) if a != b][:4]
    meta = {n: v for n, v in _kernel_meta("swx_gemm.hip").items() if "gemm_f16_big8" in n}
    assert len(meta) == 1
    for n, v in meta.items():
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_count"] <= 256, (n, v)


def test_f32_rows64_asm_loads_are_not_touched_before_their_wait():
    """gemm_f32_rows64 (round 5) keeps its operand loads in inline asm behind hand-counted waits.  Its first hardware run computed
    garbage and faulted: the idle re-loads of the last K chunk landed in registers hipcc had already handed to the epilogue's
    address arithmetic (no store follows them, so the compiler believed them dead).  scripts/isa_asm_load_audit.py walks the
    compiled kernel in program order (loop bodies twice, the pending set carried around the back edge): between an asm load and
    the counted wait that covers it, no compiler-generated instruction may read or write its destination."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_asm_load_audit", os.path.join(ROOT, "scripts", "isa_asm_load_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = _device_asm("swx_gemm.hip").split("\n")
    for key in ("gemm_f32_rows64ILi1E", "gemm_f32_rows64ILi2E"):
        body = mod.kernel_body(lines, key)
        assert body is not None, key
        viol, n_loads, n_waits = mod.audit(body)
        assert n_loads >= 15 and n_waits >= 4, (key, n_loads, n_waits)       # prologue 2 x NL + two unrolled chunks; three counted waits + the drain
        assert not viol, (key, viol[:6])
        assert not any("scratch_" in ln for ln in body), key


def _bpermutes_per_kernel(src_name):
    """(kernel symbol -> number of ds_bpermute_b32 in its body) for one csrc file"""
    out, name = {}, None
    for ln in _device_asm(src_name).split("\n"):
        m = re.match(r"^(_Z\S+):\s", ln + " ")
        if m and not ln.startswith(".L"):
            name = m.group(1)
            out.setdefault(name, 0)
        elif name and "ds_bpermute_b32" in ln:
            out[name] += 1
        if "s_endpgm" in ln:
            name = None
    return out


def test_reductions_exchange_lanes_on_the_valu_except_in_the_streaming_attention_kernels():
    """Round 6 (csrc/swx_common.h::lane_xor): every wave / block reduction, the LayerNorm statistics of the dec GEMMs and the flash
    kernels' row statistics exchange lanes with v_permlane16/32_swap and DPP -- no ds_bpermute (an LDS round trip on the dependency
    chain) -- EXCEPT the two HBM-streaming attention kernels, which keep it on purpose (measured: profiles/r06_c25_*.json): the fp16
    decode cross-attention body and the key-split form of attn_flash_f32.  A regression in either direction shows up here."""
    for src in ("swx_decstep.hip", "swx_decode.hip", "swx_norm.hip", "swx_align.hip"):
        bad = {k: v for k, v in _bpermutes_per_kernel(src).items() if v}
        assert not bad, (src, bad)
    attn = _bpermutes_per_kernel("swx_attn.hip")
    flash2 = {k: v for k, v in attn.items() if "attn_flash2_f16" in k}
    assert flash2 and not any(flash2.values()), flash2
    split = {k: v for k, v in attn.items() if "attn_flash_f32" in k and k.endswith("ELb1EEEv8AttnArgs")}
    assert len(split) == 2 and all(v == 4 for v in split.values()), split        # tmax (2) + the row sums of the merge (2)
    cross = {k: v for k, v in attn.items() if "attn_decode_cross" in k}
    assert cross and all(v >= 4 for v in cross.values()), cross                  # max + sum exchanges of its streaming loop
    step = {k: v for k, v in attn.items() if "self_attn_step_f16" in k}
    assert step and all(v <= 32 for v in step.values()), step                    # only the ancestor-id broadcasts (__shfl by lane index)
