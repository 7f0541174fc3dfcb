"""GPU parity tests proper: the HIP path (through the C ABI of libulc_amd.so) against the
oracle on the same seeded inputs.  Bit-exact on the packed stream, WindowCtrl and
BlockComplexity; MDCT coefficients additionally within 1e-5 (relative to the block
peak) of the binary64 referee via the oracle's own referee test."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd"))
from ulc_testlib import synth_pcm, oracle_encode_debug, oracle_decode_stream, synth_block_stream

pytestmark = pytest.mark.gpu


def _amd():
    import ulc_amd
    return ulc_amd


def _streams(B, nblk, bs, ch, rate, transient, seed):
    return np.stack([synth_pcm(s, nblk * bs, ch, rate, transient=transient, seed=seed) for s in range(B)])


def _compare_encode(enc_out, ref, s, k0, K, dbg=None, what=""):
    out, bits, wc, cplx = enc_out
    for k in range(K):
        kk = k0 + k
        tag = f"{what} stream {s} block {kk}"
        assert wc[s, k] == ref["wc"][kk], f"{tag}: WindowCtrl {wc[s, k]:#x} != {ref['wc'][kk]:#x}"
        if dbg is not None:
            assert np.array_equal(dbg["coef"][s, k], ref["coef"][kk]), f"{tag}: MDCT coefficients differ"
            assert np.array_equal(dbg["noise"][s, k], ref["noise"][kk]), f"{tag}: noise spectrum differs"
            assert np.array_equal(dbg["keys"][s, k], ref["keys"][kk]), f"{tag}: importance keys differ"
            assert dbg["nout"][s, k] == ref["nout"][kk], f"{tag}: nOutCoef {dbg['nout'][s, k]} != {ref['nout'][kk]}"
            keep_ref = (ref["ranks"][kk] < ref["nout"][kk]).astype(np.uint8)
            assert np.array_equal(dbg["keep"][s, k], keep_ref), f"{tag}: kept-coefficient set differs"
        assert cplx[s, k].tobytes() == ref["cplx"][kk].tobytes(), f"{tag}: BlockComplexity {cplx[s, k]} != {ref['cplx'][kk]}"
        assert bits[s, k] == ref["bits"][kk], f"{tag}: size {bits[s, k]} != {ref['bits'][kk]}"
        nb = bits[s, k] // 8
        assert np.array_equal(out[s, k, :nb], ref["out"][kk, :nb]), f"{tag}: stream bytes differ"


@pytest.mark.parametrize("bs,ch,rate,transient,q", [
    (2048, 2, 44100, True, 50.0),      # the bench shape
    (2048, 1, 44100, True, 50.0),      # BASELINE config 1 shape
    (4096, 2, 48000, True, 70.0),      # config 5 shape
    (256, 1, 44100, True, 90.0),       # smallest block: D=4..7 window codes
    (512, 3, 32000, True, 30.0),       # odd channel count (unpaired last channel)
    (2048, 2, 44100, False, 10.0),
])
def test_encode_vbr_bit_exact(bs, ch, rate, transient, q):
    amd = _amd()
    B, calls, K = 6, 3, 5
    pcm = _streams(B, calls * K, bs, ch, rate, transient, seed=bs + ch)
    enc = amd.BatchEncoder(B, ch, bs, rate, K)
    refs = [oracle_encode_debug(pcm[s], bs, rate, 0, q, slot=enc.slot) for s in range(B)]
    for c in range(calls):                      # state must carry across calls
        res = enc.encode(pcm[:, c * K * bs:(c + 1) * K * bs], amd.MODE_VBR, q)
        dbg = enc.debug_fetch()
        for s in range(B):
            _compare_encode(res, refs[s], s, c * K, K, dbg, f"bs={bs} ch={ch}")
    enc.close()


@pytest.mark.parametrize("bs,ch,rate,mode,p0,p1", [
    (2048, 2, 48000, 1, 64.0, 0.0),     # BASELINE config 4 shape: CBR 64 kbps 48 kHz stereo
    (2048, 1, 44100, 1, 32.0, 0.0),
    (1024, 2, 44100, 2, 96.0, 0.45),    # ABR
])
def test_encode_cbr_abr_bit_exact(bs, ch, rate, mode, p0, p1):
    amd = _amd()
    B, K = 4, 6
    pcm = _streams(B, K, bs, ch, rate, True, seed=17)
    enc = amd.BatchEncoder(B, ch, bs, rate, K)
    res = enc.encode(pcm, mode, p0, p1)
    dbg = enc.debug_fetch()
    for s in range(B):
        ref = oracle_encode_debug(pcm[s], bs, rate, mode, p0, p1, slot=enc.slot)
        _compare_encode(res, ref, s, 0, K, dbg, f"mode={mode}")
    enc.close()


def test_rate_search_at_high_rates_over_several_calls():
    """ABR at 229 kbps on a signal beyond full scale, three calls of ten blocks: what the randomised sweep found wrong in the
    first form of the rate search's key window (round 4: equal keys at a probe's threshold; blocks whose search is over
    before the first probe).  Sizes and bytes against the oracle."""
    amd = _amd()
    bs, ch, rate, B, K, calls = 2048, 2, 48000, 3, 10, 3
    pcm = np.stack([synth_pcm(s, calls * K * bs, ch, rate, transient=True, seed=760153175) for s in range(B)]) * np.float32(8.0)
    enc = amd.BatchEncoder(B, ch, bs, rate, K)
    outs = [enc.encode(pcm[:, c * K * bs:(c + 1) * K * bs], 2, 228.95, 0.33) for c in range(calls)]
    out = np.concatenate([o[0] for o in outs], axis=1); bits = np.concatenate([o[1] for o in outs], axis=1)
    for s in range(B):
        ref = oracle_encode_debug(pcm[s], bs, rate, 2, 228.95, 0.33, slot=enc.slot)
        assert np.array_equal(bits[s], ref["bits"]), f"stream {s} sizes"
        for k in range(calls * K):
            nb = bits[s, k] // 8
            assert np.array_equal(out[s, k, :nb], ref["out"][k, :nb]), f"stream {s} block {k} bytes"
    enc.close()


@pytest.mark.parametrize("mode,p0", [(0, 50.0), (1, 96.0)])
def test_encode_tie_straddle_uses_exact_heapsort_order(mode, p0):
    """Key ties straddling the cut need the exact heapsort emulation (SURVEY.md §7 hard part 2).
    Natural ties are rare (~4e-4 per block), so force them: a 4-channel stream whose two
    stereo pairs are identical makes every key of channel 0 tie with channel 2 (and 1 with 3);
    about every other block then has its threshold tie group straddling the cut."""
    amd = _amd()
    bs, rate, B, K = 512, 44100, 8, 12
    base = _streams(B, K, bs, 2, rate, True, seed=77)
    pcm = np.concatenate([base, base], axis=2)               # [B][n][4] = (L, R, L, R)
    enc = amd.BatchEncoder(B, 4, bs, rate, K)
    res = enc.encode(pcm, mode, p0)
    nfb = enc.last_fallbacks()
    dbg = enc.debug_fetch()
    for s in range(B):
        ref = oracle_encode_debug(pcm[s], bs, rate, mode, p0, slot=enc.slot)
        _compare_encode(res, ref, s, 0, K, dbg, "forced-ties")
    assert nfb >= B, f"only {nfb} tie-straddle blocks: the heapsort path is not covered"
    enc.close()


@pytest.mark.parametrize("mode,p0,every", [(0, 50.0, 2), (0, 50.0, 1), (1, 64.0, 2)])
def test_exact_path_at_scale_in_stereo(mode, p0, every):
    """the exact (heapsort-rank) path walks its list with a fixed grid of 128 workgroups = 256 blocks per trip;
    natural ties put ~50 blocks of the bench batch there, so nothing ever made a second trip.  The test hook hands every
    `every`-th block of a stereo BlockSize-2048 call to that path (640 blocks per call: 320 or 640 of them, several trips
    per wave; the wave writer's direct packing is a one-trip protocol and must stand back) - same bytes as the oracle."""
    amd = _amd()
    bs, ch, rate, B, K = 2048, 2, 44100, 40, 16
    pcm = _streams(B, K, bs, ch, rate, True, seed=515)
    enc = amd.BatchEncoder(B, ch, bs, rate, K)
    enc.force_exact(every)
    res = enc.encode(pcm, mode, p0)
    nfb = enc.last_fallbacks()
    assert nfb > 256, f"only {nfb} blocks on the exact path: no wave made a second trip"        # (a stream's first block is silent: nothing to select)
    for s in range(B):
        ref = oracle_encode_debug(pcm[s], bs, rate, mode, p0, slot=enc.slot)
        _compare_encode(res, ref, s, 0, K, None, f"forced exact path every={every}")
    enc.close()


@pytest.mark.parametrize("mode,p0", [(0, 50.0), (1, 64.0)])
def test_exact_path_over_several_groups_of_rank_slots(mode, p0):
    """the resident rank slots are sized by what the device has free (and halved until the allocation succeeds),
    so a call may hold fewer slots than blocks on the exact path: the launch then walks the slots group by group.  With
    ULCX_RANK_SLOTS=96 and every block of a 640-block call forced onto that path (7 groups), same bytes as the oracle."""
    amd = _amd()
    bs, ch, rate, B, K = 2048, 2, 44100, 40, 16
    pcm = _streams(B, K, bs, ch, rate, True, seed=616)
    old = os.environ.get("ULCX_RANK_SLOTS")
    os.environ["ULCX_RANK_SLOTS"] = "96"
    try:
        enc = amd.BatchEncoder(B, ch, bs, rate, K)
    finally:
        if old is None: os.environ.pop("ULCX_RANK_SLOTS", None)
        else: os.environ["ULCX_RANK_SLOTS"] = old
    enc.force_exact(1)
    res = enc.encode(pcm, mode, p0)
    assert enc.last_fallbacks() > 5 * 96
    for s in range(B):
        ref = oracle_encode_debug(pcm[s], bs, rate, mode, p0, slot=enc.slot)
        _compare_encode(res, ref, s, 0, K, None, "exact path, 96 rank slots")
    enc.close()


def _sparse_pcm(n, ch, rate, seed, floor):
    """A few steady tones over a noise floor `floor` below them: at a low VBR quality a block keeps a handful of coefficients
    and the gaps between them are hundreds to thousands of zeros - several noise runs per gap (a run code covers <= 527)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / rate
    x = np.zeros(n)
    for f, a in zip(rng.uniform(200.0, 0.45 * rate, 3), rng.uniform(0.1, 0.3, 3)):
        x += a * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28))
    pcm = np.stack([x + floor * rng.standard_normal(n) for _ in range(ch)], axis=1)
    if ch == 2:
        pcm[:, 1] = 0.7 * pcm[:, 1] + 0.3 * pcm[:, 0]
    return (np.clip(np.rint(pcm * 32768.0), -32768, 32767) / 32768.0).astype(np.float32)


@pytest.mark.parametrize("bs,ch,q,floor", [(4096, 2, 25.0, 3e-3), (8192, 1, 20.0, 1e-2), (8192, 1, 30.0, 2e-4), (4096, 2, 35.0, 4e-5), (2048, 2, 15.0, 2e-3),
                                           (16384, 1, 20.0, 1e-3), (32768, 1, 25.0, 1e-3)])
def test_long_zero_gaps_with_several_noise_runs(bs, ch, q, floor):
    """k_nsums speculates EVERY noise run of a gap (run r of the gap in front of kept coefficient i sits in component
    r & 1 of gapSum[i - (r >> 1)]) and the writer chains through them as long as each run is coded as noise; nothing
    targeted gaps of two or more runs (> 543 zeros) or three (> 1070).  Sparse spectra in large blocks: gaps of up to thousands
    of zeros, noise floors from audible to below the quantiser (a run whose level quantises to 0 is coded as zeros and breaks
    the chain).  BlockSize 16384 is the largest with speculative sums; at 32768 the writers form every sum themselves."""
    amd = _amd()
    rate, B, K = 44100, 3, (6 if bs <= 8192 else 4)
    pcm = np.stack([_sparse_pcm(K * bs, ch, rate, 900 + 7 * s, floor) for s in range(B)])
    enc = amd.BatchEncoder(B, ch, bs, rate, K)
    res = enc.encode(pcm, amd.MODE_VBR, q)
    long2 = long3 = 0
    for s in range(B):
        ref = oracle_encode_debug(pcm[s], bs, rate, 0, q, slot=enc.slot)
        _compare_encode(res, ref, s, 0, K, None, f"sparse bs={bs} q={q} floor={floor}")
        for k in range(K):
            keep = np.flatnonzero(ref["ranks"][k] < ref["nout"][k])
            for c0 in range(ch):
                kk = keep[(keep >= c0 * bs) & (keep < (c0 + 1) * bs)] - c0 * bs
                gaps = np.diff(np.concatenate([[-1], kk, [bs]])) - 1
                long2 += int((gaps > 543).sum()); long3 += int((gaps > 1070).sum())
    assert long2 > 0, "no gap of more than 543 zeros: the input is not sparse enough for this test"
    if bs >= 4096:
        assert long3 > 0, "no gap of more than 1070 zeros"
    enc.close()


def test_encode_many_blocks_bit_exact():
    """A few thousand blocks of the bench shape, every byte compared with the oracle."""
    amd = _amd()
    bs, ch, rate, B, K = 2048, 2, 44100, 96, 24
    pcm = _streams(B, K, bs, ch, rate, True, seed=2024)
    enc = amd.BatchEncoder(B, ch, bs, rate, K)
    res = enc.encode(pcm, amd.MODE_VBR, 50.0)
    for s in range(B):
        ref = oracle_encode_debug(pcm[s], bs, rate, 0, 50.0, slot=enc.slot)
        _compare_encode(res, ref, s, 0, K, None, "many-blocks")
    enc.close()


@pytest.mark.parametrize("bs,ch,rate,q", [(2048, 2, 44100, 50.0), (2048, 1, 44100, 80.0), (4096, 2, 48000, 60.0), (256, 2, 44100, 90.0)])
def test_decode_bit_exact(bs, ch, rate, q):
    amd = _amd()
    B, calls, K = 5, 2, 6
    pcm = _streams(B, calls * K, bs, ch, rate, True, seed=3)
    slot = 2 * ch * bs + 16
    refs = [oracle_encode_debug(pcm[s], bs, rate, 0, q, slot=slot) for s in range(B)]
    blocks = np.stack([r["out"] for r in refs])                  # [B][nblk][slot]
    dec = amd.BatchDecoder(B, ch, bs, K)
    got = []
    gbits = []
    for c in range(calls):
        p, b = dec.decode(blocks[:, c * K:(c + 1) * K])
        got.append(p); gbits.append(b)
    got = np.concatenate(got, axis=1); gbits = np.concatenate(gbits, axis=1)
    for s in range(B):
        rc, ref_pcm, ref_bits = oracle_decode_stream(refs[s]["out"], ch, bs)
        assert rc == 0
        assert np.array_equal(gbits[s], ref_bits), f"stream {s}: bits consumed differ"
        assert np.array_equal(got[s], ref_pcm), f"stream {s}: decoded PCM differs (max {np.abs(got[s]-ref_pcm).max()})"
    dec.close()


@pytest.mark.parametrize("bs,rate", [(2048, 44100), (4096, 48000), (1024, 44100)])
def test_decode_few_long_streams_cut_evenly(bs, rate):
    """Round 3: a batch that does not fill the machine with one workgroup per stream - here 3 stereo streams of 40 blocks per
    call - is synthesised over an even cut of its (stream, block) pairs: a workgroup that starts inside a stream rebuilds the
    lapping state from the block in front of its range, jumps the RNG, and takes LastSubBlockSize from the previous window
    code.  Two calls (the state arrays swap), window switching, a corrupt block in the middle of one stream; against the
    oracle, and against the same decoder with the cut switched off."""
    amd = _amd()
    B, calls, K, ch = 3, 2, 40, 2
    pcm = _streams(B, calls * K, bs, ch, rate, True, seed=77)
    slot = 2 * ch * bs + 16
    refs = [oracle_encode_debug(pcm[s], bs, rate, 0, 50.0, slot=slot) for s in range(B)]
    blocks = np.stack([r["out"] for r in refs]).copy()
    blocks[1, K + 11, 2:40] = 0x11                                # stream 1 dies in the second call (long zero runs overrunning the subblock)
    outs = {}
    for split in ("1", "0"):
        old = os.environ.get("ULCX_DSYN_SPLIT")
        os.environ["ULCX_DSYN_SPLIT"] = split
        try:
            dec = amd.BatchDecoder(B, ch, bs, K)
            got, gbits = [], []
            for c in range(calls):
                p, b = dec.decode(blocks[:, c * K:(c + 1) * K])
                got.append(p); gbits.append(b)
            outs[split] = (np.concatenate(got, axis=1), np.concatenate(gbits, axis=1))
            dec.close()
        finally:
            if old is None: os.environ.pop("ULCX_DSYN_SPLIT", None)
            else: os.environ["ULCX_DSYN_SPLIT"] = old
    assert np.array_equal(outs["1"][1], outs["0"][1]) and np.array_equal(outs["1"][0], outs["0"][0]), "even cut and one workgroup per stream disagree"
    for s in range(B):
        rc, ref_pcm, ref_bits = oracle_decode_stream(blocks[s], ch, bs)
        n = calls * K
        got, gb = outs["1"][0][s], outs["1"][1][s]
        if s == 1:
            assert (gb[K + 11:] == 0).all() and not got[(K + 11) * bs:].any(), "a dead stream must stay silent"
            n = K + 11
        assert np.array_equal(gb[:n], ref_bits[:n]), f"stream {s}: bits consumed differ"
        assert np.array_equal(got[:n * bs], ref_pcm[:n * bs]), f"stream {s}: decoded PCM differs"


def test_decode_last_round_cut_into_pieces():
    """Round 5: a batch of more streams than the device holds synthesis workgroups keeps one workgroup per stream for its whole
    rounds and cuts the streams of the partly empty last round into pieces of 8 blocks (ulcx_dec_tail_plan).  Whole rounds plus
    a last round two thirds full of stereo streams, 26 blocks per call (1536 resident workgroups on an MI355X: 2560 streams, 1024
    of them in 3328 pieces whose boundaries fall anywhere; the expected launch is what the exported plans say for the residency
    the decoder reports), two calls (the state arrays swap), a corrupt block inside a cut stream; against the same decoder with the cut
    switched off (every sample of every stream) and against the oracle on streams of both kinds."""
    import torch
    amd = _amd()
    sys.path.insert(0, ROOT)
    import bench
    calls, K, ch, bs, rate = 2, 26, 2, 2048, 44100
    # the batch shape follows from the device: whole rounds + a last round two thirds full, so that the tail plan is the one
    # taken whatever the residency of the synthesis kernel's instantiations is (1536 on an MI355X: B = 2560)
    import ctypes as C
    probe = amd.BatchDecoder(8, ch, bs, K)
    resident = probe.last_cut()[2]
    probe.close()
    assert resident > 0, "stereo BlockSize 2048 runs the two-wave synthesis kernel"
    B = resident + (resident * 2 // 3)
    L = amd.lib()
    want_full = C.c_int32(0)
    want_tail = L.ulcx_dec_tail_plan(B, K, resident, C.byref(want_full))
    assert L.ulcx_dec_split_plan(B, K, resident) == 0 and want_tail > 0 and want_full.value == resident, (B, resident, want_tail)
    dev = torch.device("cuda", 0)
    bench.RATE = rate
    pcm = bench.make_pcm(torch, B, calls * K * bs, dev, 4242, bursts_per_s=6.0, decades=3.0)
    enc = amd.BatchEncoder(B, ch, bs, rate, K)
    slot = enc.slot
    out = torch.zeros(B, calls * K, slot, dtype=torch.uint8, device=dev)
    for c in range(calls):
        o = torch.zeros(B, K, slot, dtype=torch.uint8, device=dev); b = torch.zeros(B, K, dtype=torch.int32, device=dev)
        enc.encode_dev(pcm[:, c * K * bs:(c + 1) * K * bs].contiguous().data_ptr(), K, o.data_ptr(), b.data_ptr(), 0, 0, mode=0, p0=50.0)
        torch.cuda.synchronize()
        out[:, c * K:(c + 1) * K] = o
    enc.close()
    dead_s, dead_k = resident + (B - resident) // 2, K + 5
    out[dead_s, dead_k, 2:40] = 0x11                                  # a stream of the cut part dies in the second call
    outs = {}
    cut_seen = False
    for tail in ("1", "0"):
        old = os.environ.get("ULCX_DSYN_TAIL")
        os.environ["ULCX_DSYN_TAIL"] = tail
        try:
            dec = amd.BatchDecoder(B, ch, bs, K)
            dp = torch.zeros(B, calls * K * bs, ch, dtype=torch.float32, device=dev); db = torch.zeros(B, calls * K, dtype=torch.int32, device=dev)
            for c in range(calls):
                p = torch.zeros(B, K * bs, ch, dtype=torch.float32, device=dev); b = torch.zeros(B, K, dtype=torch.int32, device=dev)
                dec.decode_dev(out[:, c * K:(c + 1) * K].contiguous().data_ptr(), slot, K, p.data_ptr(), b.data_ptr())
                torch.cuda.synchronize()
                dp[:, c * K * bs:(c + 1) * K * bs] = p; db[:, c * K:(c + 1) * K] = b
            outs[tail] = (dp, db)
            grid, whole, res2 = dec.last_cut()
            assert res2 == resident
            if tail == "0": assert grid == 0
            else:                                                      # what the exported plan says for this device
                assert whole == want_full.value and grid == want_full.value + want_tail, (grid, whole, resident, want_tail)
                cut_seen = True
            dec.close()
        finally:
            if old is None: os.environ.pop("ULCX_DSYN_TAIL", None)
            else: os.environ["ULCX_DSYN_TAIL"] = old
    assert cut_seen
    # the PCM16-output instantiation of the cut kernel: lrintf(clamp(y * 2^15)) of the float path's output (tools/WavIO_Helper.c:56-63)
    dec = amd.BatchDecoder(B, ch, bs, K)
    for c in range(calls):
        p16 = torch.zeros(B, K * bs, ch, dtype=torch.int16, device=dev); b = torch.zeros(B, K, dtype=torch.int32, device=dev)
        dec.decode_dev_pcm16(out[:, c * K:(c + 1) * K].contiguous().data_ptr(), slot, K, p16.data_ptr(), b.data_ptr())
        torch.cuda.synchronize()
        yf = outs["1"][0][:, c * K * bs:(c + 1) * K * bs]
        want16 = torch.clamp(torch.round(yf * 32768.0), -32768, 32767).to(torch.int16)
        assert torch.equal(p16, want16), f"call {c}: PCM16 output of the cut launch differs from lrintf(clamp(y*2^15))"
    assert dec.last_cut()[0] > 0
    dec.close()
    assert torch.equal(outs["1"][1], outs["0"][1]), "bits consumed: cut and uncut last round disagree"
    assert torch.equal(outs["1"][0].view(torch.int32), outs["0"][0].view(torch.int32)), "decoded PCM: cut and uncut last round disagree"
    for s in (0, resident - 1, resident, resident + 1, resident + (B - resident) // 3, dead_s, dead_s + 1, B - 1):
        rc, ref_pcm, ref_bits = oracle_decode_stream(out[s].cpu().numpy(), ch, bs)
        got = outs["1"][0][s].cpu().numpy(); gb = outs["1"][1][s].cpu().numpy()
        n = calls * K
        if s == dead_s:
            assert (gb[dead_k:] == 0).all() and not got[dead_k * bs:].any(), "a dead stream must stay silent"
            n = dead_k
        assert np.array_equal(gb[:n], ref_bits[:n]), f"stream {s}: bits consumed differ"
        assert np.array_equal(got[:n * bs].view(np.uint32), ref_pcm[:n * bs].view(np.uint32)), f"stream {s}: decoded PCM differs"


@pytest.mark.parametrize("bs,ch,rate,q", [(2048, 2, 44100, 50.0), (512, 1, 48000, 70.0)])
def test_decode_block_filling_its_slot_exactly(bs, ch, rate, q):
    """slotBytes = the byte count of the call's largest block (what a caller sizing slots to the container's MaxBlockSize
    passes): that block ends on the last byte of its slot and must decode like any other (no hidden headroom)."""
    amd = _amd()
    B, K = 7, 6
    pcm = _streams(B, K, bs, ch, rate, True, seed=11)
    wide = 2 * ch * bs + 16
    refs = [oracle_encode_debug(pcm[s], bs, rate, 0, q, slot=wide) for s in range(B)]
    nbytes = np.stack([(r["bits"] + 7) // 8 for r in refs])         # [B][K]
    slot = int(nbytes.max())
    tight = np.zeros((B, K, slot), np.uint8)
    for s in range(B):
        for k in range(K):
            tight[s, k, :nbytes[s, k]] = refs[s]["out"][k, :nbytes[s, k]]
    dec = amd.BatchDecoder(B, ch, bs, K)
    got, gbits = dec.decode(tight)
    for s in range(B):
        rc, ref_pcm, ref_bits = oracle_decode_stream(refs[s]["out"], ch, bs)
        assert rc == 0
        assert np.array_equal(gbits[s], ref_bits), f"stream {s}: bits consumed differ with slot = {slot} bytes"
        assert np.array_equal(got[s], ref_pcm), f"stream {s}: decoded PCM differs with slot = {slot} bytes"
    dec.close()


@pytest.mark.parametrize("bs,ch,B,K,calls", [(512, 2, 40, 6, 2), (2048, 2, 70, 5, 2), (1024, 1, 33, 7, 1), (256, 3, 20, 4, 2), (4096, 2, 9, 3, 1)])
def test_decode_hand_assembled_streams(bs, ch, B, K, calls):
    """Every code of the block syntax, including those the encoder never emits: decimation codes
    2h-7h, all overlap scales, extended quantizers, long zero runs, both stop codes, units that open with a stop."""
    amd = _amd()
    slot = 2 * ch * bs + 16
    streams = [synth_block_stream(1000 + 17 * s + bs, calls * K, ch, bs, slot) for s in range(B)]
    blocks = np.stack([st[0] for st in streams])
    dec = amd.BatchDecoder(B, ch, bs, K)
    got, gbits = [], []
    for c in range(calls):
        p, b = dec.decode(blocks[:, c * K:(c + 1) * K])
        got.append(p); gbits.append(b)
    got = np.concatenate(got, axis=1); gbits = np.concatenate(gbits, axis=1)
    for s in range(B):
        rc, ref_pcm, ref_bits = oracle_decode_stream(blocks[s], ch, bs)
        assert rc == 0, f"oracle rejected hand-assembled stream {s} at block {rc - 1}"
        assert np.array_equal(ref_bits, streams[s][1]), f"stream {s}: oracle consumed {ref_bits} bits, assembled {streams[s][1]}"
        assert np.array_equal(gbits[s], ref_bits), f"stream {s}: bits consumed differ: {gbits[s]} vs {ref_bits}"
        assert np.array_equal(got[s].view(np.uint32), ref_pcm.view(np.uint32)), \
            f"stream {s}: decoded PCM differs (max {np.abs(got[s]-ref_pcm).max()})"
    dec.close()


def test_roundtrip_delay_and_snr_at_bench_shape():
    """Size-independent property: encode -> decode reproduces the input delayed by exactly
    2*BlockSize samples with codec-level SNR."""
    amd = _amd()
    bs, ch, rate, B, K = 2048, 2, 44100, 64, 16
    pcm = _streams(B, K, bs, ch, rate, True, seed=99)
    enc = amd.BatchEncoder(B, ch, bs, rate, K)
    out, bits, wc, cplx = enc.encode(pcm, amd.MODE_VBR, 60.0)
    dec = amd.BatchDecoder(B, ch, bs, K)
    got, gbits = dec.decode(out)
    assert (gbits > 0).all() and (gbits <= bits).all() and (bits - gbits < 8).all()
    d = 2 * bs
    x, y = pcm[:, :-d], got[:, d:]
    snr = 10 * np.log10((x ** 2).sum() / ((x - y) ** 2).sum())
    assert snr > 12.0, snr
    enc.close(); dec.close()


def test_decode_corrupt_blocks_are_rejected_without_hanging():
    """Corrupt input must come back as 'bits consumed = 0' (ulcDecoder.c:127,139,154) and kill
    only its own stream; a stream of endless quantizer changes (no coefficient ever produced)
    must not walk off the slot."""
    amd = _amd()
    bs, ch, rate, B, K = 2048, 2, 44100, 6, 4
    pcm = _streams(B, K, bs, ch, rate, True, seed=8)
    slot = 2 * ch * bs + 16
    refs = [oracle_encode_debug(pcm[s], bs, rate, 0, 50.0, slot=slot) for s in range(B)]
    blocks = np.stack([r["out"] for r in refs]).copy()
    blocks[1, 2, :] = 0x0F                              # Fh,0h forever: quantizer changes only
    blocks[3, 1, 2:40] = 0x11                           # long zero runs overrunning the subblock
    blocks[4, 0, :] = 0xFF
    dec = amd.BatchDecoder(B, ch, bs, K)
    got, gbits = dec.decode(blocks)
    assert (gbits[1, :2] > 0).all() and (gbits[1, 2:] == 0).all()
    assert gbits[3, 0] > 0 and (gbits[3, 1:] == 0).all()
    for s in (0, 2, 5):                                 # untouched streams are still bit-exact
        rc, ref_pcm, ref_bits = oracle_decode_stream(refs[s]["out"], ch, bs)
        assert np.array_equal(gbits[s], ref_bits) and np.array_equal(got[s], ref_pcm)
    assert np.isfinite(got).all()
    dec.close()


def test_packed_stream_pack_and_decode():
    """.ulc payloads: blocks back to back, each rounded up to a byte, no lengths.
    GPU pack == host concatenation; GPU packed decode (two calls, position carried) == oracle decode."""
    import ctypes as C
    import torch
    amd = _amd()
    bs, ch, rate, B, K = 1024, 2, 44100, 5, 8
    pcm = _streams(B, K, bs, ch, rate, True, seed=12)
    slot = 2 * ch * bs + 16
    refs = [oracle_encode_debug(pcm[s], bs, rate, 0, 60.0, slot=slot) for s in range(B)]
    payloads = [b"".join(r["out"][k, : (r["bits"][k] + 7) // 8].tobytes() for k in range(K)) for r in refs]
    stride = max(len(p) for p in payloads) + 64
    host = np.zeros((B, stride), np.uint8)
    for s, p in enumerate(payloads):
        host[s, : len(p)] = np.frombuffer(p, np.uint8)
    nbytes = np.array([len(p) for p in payloads], np.int32)
    # pack kernel
    dev = torch.device("cuda", 0)
    slots = torch.from_numpy(np.stack([r["out"] for r in refs])).to(dev)
    bits = torch.from_numpy(np.stack([r["bits"] for r in refs])).to(dev)
    pay = torch.zeros(B, stride, dtype=torch.uint8, device=dev); pb = torch.zeros(B, dtype=torch.int32, device=dev); mb = torch.zeros(B, dtype=torch.int32, device=dev)
    rc = amd.lib().ulcx_pack_streams_dev(0, B, K, slot, slots.data_ptr(), bits.data_ptr(), pay.data_ptr(), stride, pb.data_ptr(), mb.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(pb.cpu().numpy(), nbytes) and np.array_equal(pay.cpu().numpy(), host)
    assert np.array_equal(mb.cpu().numpy(), np.array([((r["bits"] + 7) // 8).max() for r in refs]))
    # packed decode in two calls
    dec = amd.BatchDecoder(B, ch, bs, K // 2)
    p1, b1 = dec.decode_packed(host, nbytes, K // 2)
    p2, b2 = dec.decode_packed(host, nbytes, K // 2)
    got = np.concatenate([p1, p2], axis=1); gb = np.concatenate([b1, b2], axis=1)
    for s in range(B):
        rc2, ref_pcm, ref_bits = oracle_decode_stream(refs[s]["out"], ch, bs)
        assert np.array_equal(gb[s], ref_bits) and np.array_equal(got[s], ref_pcm), s
    # the same from a payload uploaded once (what ulcx-tool does for whole files)
    dec.upload_payload(host, nbytes)
    p5, b5 = dec.decode_resident(K // 2)
    p6, b6 = dec.decode_resident(K // 2)
    assert np.array_equal(np.concatenate([p5, p6], axis=1), got) and np.array_equal(np.concatenate([b5, b6], axis=1), gb)
    # a truncated payload ends its stream there
    dec.reset()
    cut = nbytes.copy(); cut[2] = nbytes[2] // 2
    p3, b3 = dec.decode_packed(host, cut, K // 2)
    p4, b4 = dec.decode_packed(host, cut, K // 2)
    b34 = np.concatenate([b3, b4], axis=1)
    assert (b34[2] == 0).any() and (b34[[0, 1, 3, 4]] > 0).all()
    dec.close()


@pytest.mark.parametrize("bs,ch,rate,B,K,calls,q", [
    (8192, 2, 48000, 2, 3, 1, 40.0),     # largest block size the fast transform kernel takes
    (16384, 2, 48000, 1, 3, 1, 50.0),    # above it: one array at a time (k_xf_big), general decoder kernel
    (32768, 1, 44100, 2, 3, 1, 50.0),    # the reference's largest BlockSize (ulcEncoder.c:32-34)
    (8192, 6, 48000, 1, 2, 1, 50.0),     # three M/S pairs of large blocks
    (2048, 6, 48000, 2, 4, 1, 50.0),     # 5.1-style: three M/S pairs
    (1024, 5, 44100, 3, 4, 1, 70.0),     # odd channel count > 2
    (2048, 2, 44100, 1, 1, 6, 50.0),     # one stream, one block per call: the drop-in shim's shape
    (256, 2, 8000, 3, 9, 3, 95.0),       # smallest block, low rate, high quality (dense coding)
])
def test_encode_decode_unusual_geometries(bs, ch, rate, B, K, calls, q):
    amd = _amd()
    pcm = _streams(B, K * calls, bs, ch, rate, True, seed=bs + 7 * ch)
    enc = amd.BatchEncoder(B, ch, bs, rate, K)
    refs = [oracle_encode_debug(pcm[s], bs, rate, 0, q, slot=enc.slot) for s in range(B)]
    outs = []
    for c in range(calls):
        res = enc.encode(pcm[:, c * K * bs:(c + 1) * K * bs], amd.MODE_VBR, q)
        for s in range(B):
            _compare_encode(res, refs[s], s, c * K, K, None, f"geom bs={bs} ch={ch}")
        outs.append(res[0])
    enc.close()
    blocks = np.concatenate(outs, axis=1)
    try:
        dec = amd.BatchDecoder(B, ch, bs, K)
    except amd.UlcError as e:
        assert "LDS" in str(e) or "not built" in str(e), e          # documented limit
        return
    got = np.concatenate([dec.decode(blocks[:, c * K:(c + 1) * K])[0] for c in range(calls)], axis=1)
    for s in range(B):
        rc, ref_pcm, _ = oracle_decode_stream(refs[s]["out"], ch, bs)
        assert rc == 0 and np.array_equal(got[s], ref_pcm), f"geom bs={bs} ch={ch} stream {s}"
    dec.close()


@pytest.mark.parametrize("bs,ch,rate,calls,mode,p0", [
    (2048, 2, 44100, (3, 1, 4), 0, 50.0),      # stereo fast path; history crossing calls of 1 and of several blocks
    (2048, 1, 44100, (2, 2), 0, 50.0),
    (512, 3, 32000, (1, 1, 3), 0, 40.0),       # odd channel count: scalar loads
    (2048, 2, 48000, (3, 2), 1, 64.0),         # CBR
])
def test_pcm16_ingest_and_output_match_the_wav_io_conversions(bs, ch, rate, calls, mode, p0):
    """SURVEY.md §8f rank 4.  PCM16 ingest must give the stream the float path gives for x * 2^-15
    (tools/WavIO_Helper.c:49-55) - checked against the oracle too - and PCM16 output must equal
    lrintf(clamp(y * 2^15)) of the float path's output (tools/WavIO_Helper.c:56-63)."""
    import torch
    amd = _amd()
    B, Kmax = 5, max(calls)
    nblk = sum(calls)
    x = _streams(B, nblk, bs, ch, rate, True, seed=31)
    x16 = np.clip(np.rint(x * 32767.0), -32768, 32767).astype(np.int16)
    xf = x16.astype(np.float32) * np.float32(2.0 ** -15)
    modes = (amd.MODE_VBR, amd.MODE_CBR)
    encF = amd.BatchEncoder(B, ch, bs, rate, Kmax)
    encS = amd.BatchEncoder(B, ch, bs, rate, Kmax)
    decF = amd.BatchDecoder(B, ch, bs, Kmax)
    decS = amd.BatchDecoder(B, ch, bs, Kmax)
    slot = encS.slot
    refs = [oracle_encode_debug(xf[s], bs, rate, mode, p0, slot=slot) for s in range(2)]
    k0 = 0
    for K in calls:
        seg16 = np.ascontiguousarray(x16[:, k0 * bs:(k0 + K) * bs])
        want = encF.encode(xf[:, k0 * bs:(k0 + K) * bs], modes[mode], p0)
        d_in = torch.from_numpy(seg16).cuda()
        d_out = torch.zeros((B, K, slot), dtype=torch.uint8, device="cuda")
        d_bits = torch.zeros((B, K), dtype=torch.int32, device="cuda")
        d_wc = torch.zeros((B, K), dtype=torch.int32, device="cuda")
        d_cplx = torch.zeros((B, K), dtype=torch.float32, device="cuda")
        encS.encode_dev_pcm16(d_in.data_ptr(), K, d_out.data_ptr(), d_bits.data_ptr(), d_wc.data_ptr(), d_cplx.data_ptr(),
                              mode=modes[mode], p0=p0)
        torch.cuda.synchronize()
        got = (d_out.cpu().numpy(), d_bits.cpu().numpy(), d_wc.cpu().numpy(), d_cplx.cpu().numpy())
        for a, b, name in zip(got[1:], want[1:], ("bits", "WindowCtrl", "BlockComplexity")):
            assert a.tobytes() == b.tobytes(), f"call at block {k0}: PCM16 ingest {name} differ from the float path"
        for s in range(B):
            for k in range(K):
                nb = got[1][s, k] // 8                      # (bytes of a slot past the block are not defined)
                assert np.array_equal(got[0][s, k, :nb], want[0][s, k, :nb]), f"stream {s} block {k0 + k}: PCM16 ingest bytes differ from the float path"
        for s in range(2):
            _compare_encode(got, refs[s], s, k0, K, what="pcm16")
        # decoder: float output vs PCM16 output of the same blocks
        yf, ybits = decF.decode(want[0])
        d_pcm16 = torch.zeros((B, K * bs, ch), dtype=torch.int16, device="cuda")
        d_dbits = torch.zeros((B, K), dtype=torch.int32, device="cuda")
        decS.decode_dev_pcm16(d_out.data_ptr(), slot, K, d_pcm16.data_ptr(), d_dbits.data_ptr())
        torch.cuda.synchronize()
        want16 = np.clip(np.rint(yf * np.float32(32768.0)), -32768, 32767).astype(np.int16)
        assert np.array_equal(d_dbits.cpu().numpy(), ybits)
        assert np.array_equal(d_pcm16.cpu().numpy(), want16), f"call at block {k0}: PCM16 output differs from lrintf(clamp(y*2^15))"
        k0 += K
    for o in (encF, encS, decF, decS):
        o.close()


@pytest.mark.parametrize("env", [
    {"ULCX_ASYNC_FB": "0"},                        # no side streams at all (the mode the per-kernel profiles use)
    {"ULCX_WC_PIPE": "1"},                         # window control not pipelined with the transform
    {"ULCX_WC_PIPE": "3"}, {"ULCX_WC_PIPE": "8"},
    {"ULCX_WC_STEPS": "4"},
    {"ULCX_DIRECT_PACK": "0"},                     # every block packed by k_pack (default: the wave writer packs stereo un-decimated blocks itself)
    {"ULCX_ASYNC_FB": "0", "ULCX_WC_PIPE": "1"},
    # (round 5: ULCX_WC_FUSE / ULCX_WAVE / ULCX_GAPSUMS / ULCX_BARK_UNIFORM are gone - the kernels they selected are what mono /
    #  multichannel streams, blocks beyond the wave writer's capacities, C x BlockSize > 16384 and geometries with deep Bark rings
    #  run anyway: test_encode_bit_exact's geometries, test_large_blocks..., tools/fuzz_big.py)
])
def test_runtime_switches_keep_parity(env):
    """Every launch-structure switch of DESIGN.md §8 (read from the environment when the codec objects are created and
    at each call) must leave the results bit-exact: VBR and CBR over 16-block calls (long enough for the chunked
    window-control pipeline), encode vs the oracle and decode of the result vs the oracle decoder."""
    amd = _amd()
    bs, ch, rate, B, K = 1024, 2, 44100, 5, 16
    calls = 2
    pcm = _streams(B, calls * K, bs, ch, rate, True, seed=808)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        for mode, p0 in ((amd.MODE_VBR, 45.0), (amd.MODE_CBR, 96.0)):
            enc = amd.BatchEncoder(B, ch, bs, rate, K)
            dec = amd.BatchDecoder(B, ch, bs, K)
            refs = [oracle_encode_debug(pcm[s], bs, rate, 0 if mode == amd.MODE_VBR else 1, p0, slot=enc.slot) for s in range(B)]
            outs = []
            for call in range(calls):
                res = enc.encode(pcm[:, call * K * bs:(call + 1) * K * bs], mode, p0)
                for s in range(B):
                    _compare_encode(res, refs[s], s, call * K, K, None, f"{env}")
                got, gbits = dec.decode(res[0])
                outs.append((got, gbits))
            for s in range(2):
                rc, ref_pcm, ref_bits = oracle_decode_stream(refs[s]["out"], ch, bs)
                assert rc == 0
                got = np.concatenate([o[0][s] for o in outs])
                assert np.array_equal(got.view(np.uint32), ref_pcm.view(np.uint32)), f"{env}: decoded PCM differs (stream {s})"
            enc.close(); dec.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

@pytest.mark.parametrize("pair", ["1", "0"])
def test_blocksize_4096_selection_with_one_or_two_waves_per_block(pair):
    """Stereo BlockSize 4096 selects with a wave per channel (k_select_pair, round 4; ULCX_SEL_PAIR=0: one wave holding all
    8192 keys): VBR and CBR, two calls, streams and sizes against the oracle."""
    amd = _amd()
    bs, ch, rate, B, K, calls = 4096, 2, 48000, 3, 5, 2
    pcm = _streams(B, calls * K, bs, ch, rate, True, seed=4096)
    old = os.environ.get("ULCX_SEL_PAIR")
    os.environ["ULCX_SEL_PAIR"] = pair
    try:
        for mode, p0 in ((amd.MODE_VBR, 55.0), (amd.MODE_CBR, 80.0)):
            enc = amd.BatchEncoder(B, ch, bs, rate, K)
            refs = [oracle_encode_debug(pcm[s], bs, rate, 0 if mode == amd.MODE_VBR else 1, p0, slot=enc.slot) for s in range(B)]
            for call in range(calls):
                res = enc.encode(pcm[:, call * K * bs:(call + 1) * K * bs], mode, p0)
                for s in range(B):
                    _compare_encode(res, refs[s], s, call * K, K, None, f"pair={pair} mode={mode}")
            enc.close()
    finally:
        if old is None: os.environ.pop("ULCX_SEL_PAIR", None)
        else: os.environ["ULCX_SEL_PAIR"] = old


@pytest.mark.parametrize("env,B,K", [
    ({}, 6, 16),
    ({"ULCX_WC_STEPS": "8"}, 6, 16),
    ({"ULCX_WC_PIPE": "1"}, 6, 16),
    ({}, 300, 8),
    ({"ULCX_DSYN_SPLIT": "0"}, 6, 16),
])
def test_headline_geometry_schedules_keep_parity(env, B, K):
    """The launch structures of the headline geometry - stereo, BlockSize 2048, where the transform's steady-state path
    exists: encode vs the oracle over two calls (state carry, window switching), decode vs the oracle."""
    amd = _amd()
    bs, ch, rate = 2048, 2, 44100
    pcm = _streams(B, 2 * K, bs, ch, rate, True, seed=4242)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        enc = amd.BatchEncoder(B, ch, bs, rate, K)
        dec = amd.BatchDecoder(B, ch, bs, K)
        check = range(B) if B <= 8 else (0, 1, 63, 64, 65, 150, B - 2, B - 1)
        refs = {s: oracle_encode_debug(pcm[s], bs, rate, 0, 50.0, slot=enc.slot) for s in check}
        outs = []
        for call in range(2):
            res = enc.encode(pcm[:, call * K * bs:(call + 1) * K * bs], amd.MODE_VBR, 50.0)
            for s in check:
                _compare_encode(res, refs[s], s, call * K, K, None, f"{env}")
            outs.append(dec.decode(res[0]))
        for s in check:
            rc, ref_pcm, ref_bits = oracle_decode_stream(refs[s]["out"], ch, bs)
            assert rc == 0
            got = np.concatenate([o[0][s] for o in outs])
            assert np.array_equal(got.view(np.uint32), ref_pcm.view(np.uint32)), f"{env}: decoded PCM differs (stream {s})"
        enc.close(); dec.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v



def test_two_host_threads_drive_two_encoders_and_two_decoders():
    """"one host thread (or one HIP stream) per device".  Two host threads of ONE process, each with its
    own encoder and decoder (both on device 0 of the 1-GPU box; on a multi-GPU node thread t takes device t), encode and
    decode their own half of the streams at the same time, several calls each: every byte and every sample against the
    oracle.  (ctypes releases the GIL for the duration of a library call: the calls do overlap.)"""
    import threading
    amd = _amd()
    ndev = amd.lib().ulcx_device_count()
    bs, ch, rate, B, K, calls = 2048, 2, 44100, 10, 6, 3
    pcm = _streams(2 * B, calls * K, bs, ch, rate, True, seed=909)
    slot = 2 * ch * bs + 16
    res = [None, None]

    def work(t):
        try:
            mine = pcm[t * B:(t + 1) * B]
            enc = amd.BatchEncoder(B, ch, bs, rate, K, device=t % ndev)
            dec = amd.BatchDecoder(B, ch, bs, K, device=t % ndev)
            outs, pcms = [], []
            for c in range(calls):
                o = enc.encode(mine[:, c * K * bs:(c + 1) * K * bs], amd.MODE_VBR if t == 0 else amd.MODE_CBR, 50.0 if t == 0 else 72.0)
                outs.append(o)
                pcms.append(dec.decode(o[0]))
            enc.close(); dec.close()
            res[t] = (outs, pcms)
        except Exception as e:                                   # (surfaces in the assert below)
            res[t] = e

    ths = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in ths: th.start()
    for th in ths: th.join()
    for t in range(2):
        assert not isinstance(res[t], Exception), f"thread {t}: {res[t]}"
        outs, pcms = res[t]
        for s in range(B):
            ref = oracle_encode_debug(pcm[t * B + s], bs, rate, 0 if t == 0 else 1, 50.0 if t == 0 else 72.0, slot=slot)
            for c in range(calls):
                _compare_encode(outs[c], ref, s, c * K, K, None, f"thread {t}")
            rc, ref_pcm, ref_bits = oracle_decode_stream(ref["out"], ch, bs)
            got = np.concatenate([p[0][s] for p in pcms])
            assert rc == 0 and np.array_equal(got.view(np.uint32), ref_pcm.view(np.uint32)), f"thread {t} stream {s}: decoded PCM differs"


def test_timing_events_can_be_switched_off():
    """ulcx_encoder_set_timing / ulcx_decoder_set_timing: without the per-kernel events the results are the same and
    the stage tables are empty; switched on again they are filled."""
    amd = _amd()
    bs, ch, rate, B, K = 1024, 2, 44100, 3, 4
    pcm = _streams(B, K, bs, ch, rate, True, seed=31)
    refs = [oracle_encode_debug(pcm[s], bs, rate, 0, 55.0, slot=2 * ch * bs + 16) for s in range(B)]
    for timing in (False, True):
        enc = amd.BatchEncoder(B, ch, bs, rate, K); dec = amd.BatchDecoder(B, ch, bs, K)
        enc.set_timing(timing); dec.set_timing(timing)
        res = enc.encode(pcm, amd.MODE_VBR, 55.0)
        for s in range(B):
            _compare_encode(res, refs[s], s, 0, K, None, f"timing={timing}")
        got, gbits = dec.decode(res[0])
        rc, ref_pcm, ref_bits = oracle_decode_stream(refs[0]["out"], ch, bs)
        assert rc == 0 and np.array_equal(got[0], ref_pcm) and np.array_equal(gbits[0], ref_bits)
        assert bool(enc.stage_ms()) == timing and bool(dec.stage_ms()) == timing
        enc.close(); dec.close()


def test_single_block_and_batched_calls_mixed_on_one_decoder_keep_the_noise_chain():
    """ulcx_decode_block1 uploads its host copy of the noise generator's state in front of every block; a batched call on
    the same one-stream decoder advances only the device word.  Mixed use must still draw ONE chain (ulcDecoder.c:75-81):
    blocks decoded alternately through ulcx_decode_host and ulcx_decode_block1 equal the oracle's continuous decode."""
    import ctypes as C
    import ulc_amd as amd
    bs, ch, rate, nblk = 2048, 2, 44100, 12
    pcm = synth_pcm(31, nblk * bs, ch, rate, transient=True, seed=77)
    slot = 2 * ch * bs + 16
    ref = oracle_encode_debug(pcm, bs, rate, 0, 30.0, slot=slot)           # low quality: long noise-filled gaps, many draws
    rc, want, want_bits = oracle_decode_stream(ref["out"], ch, bs)
    assert rc == 0
    dec = amd.BatchDecoder(1, ch, bs, 1)
    L = amd.lib()
    L.ulcx_decode_block1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    got = np.zeros((nblk, bs, ch), np.float32)
    for k in range(nblk):
        blk = np.ascontiguousarray(ref["out"][k])
        if (k // 3) % 2 == 0:                                              # three blocks through the batched entry ...
            p, b = dec.decode(blk[None, None, :])
            got[k] = p[0]; bits = int(b[0, 0])
        else:                                                              # ... three through the single-block one
            o = np.zeros((bs, ch), np.float32); b = C.c_int32(0); ls = C.c_int32(0)
            r = L.ulcx_decode_block1(dec.h, blk.ctypes.data, int(slot), o.ctypes.data, C.byref(b), C.byref(ls))
            assert r == 0, amd.lib().ulcx_last_error()
            got[k] = o; bits = b.value
        assert bits == want_bits[k], k
    assert np.array_equal(got.reshape(-1, ch).view(np.uint32), want.view(np.uint32)), "the noise chain was rewound or forked"
    dec.close()


@pytest.mark.parametrize("bs,ch,rate", [(2048, 2, 44100), (4096, 2, 44100), (2048, 1, 48000), (1024, 2, 32000), (8192, 1, 44100)])
def test_selection_bracket_on_degenerate_key_distributions(bs, ch, rate):
    """Round 6: the selection brackets its threshold from 128 sampled keys before it searches (k_select_wave / k_select_pair,
    BlockTransform.c:20-77 decides the same set by a full sort).  Key distributions that starve or mislead the sample: a silent
    channel (half the keys -inf), two pure tones (a handful of large keys over a floor of ties), one impulse per block (flat
    spectrum: every key close to the next), near-silence, clipping square wave, white noise at quality 100 / 1 (nearly all / nearly
    none kept) - kept sets, keys and bytes must equal the oracle's, VBR and the first probe of a rate search."""
    amd = _amd()
    K = 4
    n = K * bs
    rng = np.random.default_rng(bs + ch)
    t = np.arange(n) / rate
    sig = []
    x = synth_pcm(1, n, ch, rate, transient=True, seed=5).copy(); x[:, -1] = 0.0; sig.append(x)                       # a silent channel
    x = np.zeros((n, ch), np.float32); x[:, 0] = 0.4 * np.sin(2 * np.pi * 1000 * t) + 0.3 * np.sin(2 * np.pi * 5000 * t); x[:, -1] = x[:, 0]; sig.append(x)
    x = np.zeros((n, ch), np.float32); x[bs // 3::bs, :] = 0.9; sig.append(x)                                           # an impulse per block
    sig.append((synth_pcm(2, n, ch, rate, transient=False, seed=6) * np.float32(2.0 ** -14)).astype(np.float32))        # near silence
    x = np.sign(np.sin(2 * np.pi * 440 * t)).astype(np.float32)[:, None].repeat(ch, 1) * np.float32(1.5); sig.append(x)  # beyond full scale, clipped shape
    sig.append(rng.uniform(-1, 1, (n, ch)).astype(np.float32))                                                          # white noise
    sig = [np.ascontiguousarray(np.round(s * 32768.0) / 32768.0, dtype=np.float32) for s in sig]
    pcm = np.stack(sig)
    B = len(sig)
    for mode, p0 in ((amd.MODE_VBR, 100.0), (amd.MODE_VBR, 50.0), (amd.MODE_VBR, 1.0), (amd.MODE_CBR, 96.0)):
        enc = amd.BatchEncoder(B, ch, bs, rate, K)
        res = enc.encode(pcm, mode, p0)
        dbg = enc.debug_fetch(K)
        for s in range(B):
            ref = oracle_encode_debug(pcm[s], bs, rate, mode, p0, slot=enc.slot)
            _compare_encode(res, ref, s, 0, K, dbg if mode == amd.MODE_VBR else None, what=f"mode {mode} p0 {p0} signal {s}")
        enc.close()
