"""Values, not only bit counts: the oracle decoder's dequantised coefficients (CoefDbg) against a decoder written from
FormatSpecs.md alone (tests/spec_decoder.py) — on hand-assembled streams that use every code of the syntax and on the
oracle encoder's own output.  Noise coefficients are compared by magnitude (the spec leaves the generator open); their
signs are checked to be exactly the xorshift sequence the reference uses (ulcDecoder.c:75-81), seed 1234567 per stream."""
import numpy as np
import pytest
from ulc_testlib import synth_block_stream, synth_pcm, oracle_decode_stream_coefs, oracle_encode_stream, oracle
from spec_decoder import decode_block_coefficients


def _check_stream(blocks, bits_expected, ch, bs):
    rc, pcm, bits, coefs = oracle_decode_stream_coefs(blocks, ch, bs)
    assert rc == 0
    if bits_expected is not None:
        assert np.array_equal(bits, bits_expected)
    lib = oracle()
    seed = 1234567
    n_noise = n_plain = 0
    for k in range(blocks.shape[0]):
        spec, run_of, nyb = decode_block_coefficients(blocks[k], ch, bs)
        is_noise = run_of > 0
        assert 4 * nyb == bits[k], f"block {k}: spec decoder consumed {4 * nyb} bits, oracle {bits[k]}"
        got = coefs[k].reshape(ch, bs)
        plain = ~is_noise
        assert np.array_equal(got[plain].view(np.uint32), spec[plain].view(np.uint32)), f"block {k}: coefficient values differ from the spec"
        assert np.array_equal(np.abs(got[is_noise]).view(np.uint32), spec[is_noise].view(np.uint32)), f"block {k}: noise magnitudes differ from the spec"
        # signs: one draw per noise coefficient in coding order; inside one noise code the sign starts positive and flips
        # (cumulatively) on every draw whose top bit is set (ulcDecoder.c:146-160,166-184)
        flat_run = run_of.reshape(-1); flat_got = got.reshape(-1)
        idx = np.flatnonzero(flat_run)
        run_sign, cur = 1.0, 0
        for i in idx:
            if flat_run[i] != cur:
                cur = flat_run[i]; run_sign = 1.0
            seed = lib.orc_xorshift32(seed)
            if seed & 0x80000000:
                run_sign = -run_sign
            if flat_got[i] != 0.0:
                assert np.sign(flat_got[i]) == run_sign, f"block {k} coefficient {i}: noise sign is not the reference's xorshift sequence"
        n_noise += idx.size; n_plain += int(np.count_nonzero(got[plain]))
    return n_plain, n_noise


@pytest.mark.parametrize("bs,ch,nblk", [(512, 2, 12), (2048, 2, 6), (1024, 1, 8), (256, 3, 10)])
def test_hand_assembled_streams_decode_to_the_spec_values(bs, ch, nblk):
    slot = 2 * ch * bs + 16
    tot_p = tot_n = 0
    for s in range(6):
        blocks, bits = synth_block_stream(4242 + 31 * s + bs, nblk, ch, bs, slot, spec_only=True)
        p, n = _check_stream(blocks, bits, ch, bs)
        tot_p += p; tot_n += n
    assert tot_p > 100 and tot_n > 100


@pytest.mark.parametrize("bs,ch,rate,kw", [(2048, 2, 44100, dict(quality=50.0)), (2048, 1, 44100, dict(quality=50.0)),
                                          (4096, 2, 48000, dict(quality=70.0)), (512, 2, 32000, dict(kbps=48.0))])
def test_encoder_output_decodes_to_the_spec_values(bs, ch, rate, kw):
    pcm = synth_pcm(5, 10 * bs, ch, rate, transient=True, seed=bs + ch)
    out, bits, wc, cplx = oracle_encode_stream(pcm, bs, rate, **kw)
    p, n = _check_stream(out, None, ch, bs)
    assert p > 200 and n > 200
