#!/usr/bin/env python3
"""Randomised parity sweep (GPU vs oracle): geometry, rate-control mode, quality, signal type, call pattern.
tests/test_gpu_fuzz.py runs a bounded sweep of it (about a minute); longer runs on the GPU box:  python tools/fuzz_parity.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ulc_amd
from ulc_testlib import synth_pcm, oracle_encode_debug, oracle_decode_stream

def run(budget=120.0, seed=1, max_bs=8192, sizes=None, chans=None, rate_search=False):
  """rate_search: only CBR / ABR at high rates over >= 3 calls, half of the signals beyond full scale - the corner where the two
  faults of the rate-search window (round 4, e9a701a) lived."""
  rng = np.random.default_rng(seed)
  t0 = time.time(); n = 0; nblk = 0
  while time.time() - t0 < budget:
      bs = int(rng.choice(sizes if sizes else [b for b in [256, 512, 1024, 2048, 2048, 4096, 8192] if b <= max_bs]))
      ch = int(rng.choice(chans if chans else [1, 2, 2, 2, 3]))
      rate = int(rng.choice([22050, 32000, 44100, 48000, 96000]))
      B = int(rng.integers(1, 9)); K = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 10, 12, 16, 21])); calls = int(rng.integers(1, 4))
      if K > 8: B = min(B, 3)                                         # (long calls exercise the chunked window-control pipeline)
      mode = int(rng.choice([0, 0, 1, 2]))
      p0 = float(rng.uniform(1, 100)) if mode == 0 else float(rng.choice([rng.uniform(1, 16), rng.uniform(16, 192), rng.uniform(192, 700)]))
      if rate_search:
          mode = int(rng.choice([1, 1, 2])); p0 = float(rng.uniform(150, 700)); calls = int(rng.integers(3, 6)); K = int(rng.choice([1, 2, 3, 5, 8])); B = min(B, 4)
      p1 = float(rng.uniform(0.2, 0.9)) if mode == 2 else 0.0
      transient = bool(rng.integers(0, 2))
      amp = float(rng.choice([1.0, 1.0, 0.05, 1e-4, 0.0]))           # loud, quiet, near-silent, digital silence
      seed = int(rng.integers(0, 1 << 30))
      pcm = np.stack([synth_pcm(s, calls * K * bs, ch, rate, transient=transient, seed=seed) for s in range(B)]) * np.float32(amp)
      kind = int(rng.integers(0, 8))                                 # other signal shapes on top of the synthetic mix
      if rate_search: kind = int(rng.choice([0, 1, 6, 6, 6, 7]))
      nT = calls * K * bs
      if kind == 1:   pcm = (rng.random(pcm.shape, dtype=np.float32) * 2 - 1) * np.float32(amp if amp else 1.0)          # white noise
      elif kind == 2: pcm = pcm + np.float32(0.25)                                                                       # DC offset
      elif kind == 3: pcm = np.zeros_like(pcm); pcm[:, ::int(rng.integers(50, 3000))] = np.float32(0.9)                   # impulse train
      elif kind == 4: pcm[:, : nT // 2] = 0                                                                              # silence then signal
      elif kind == 5: pcm = np.sign(pcm).astype(np.float32) * np.float32(min(amp, 1.0) if amp else 0.5)                  # square-ish / clipped
      elif kind == 6: pcm = pcm * np.float32(8.0)                                                                        # beyond full scale
      elif kind == 7: pcm = np.round(pcm * 32767).astype(np.int16).astype(np.float32) * np.float32(2.0 ** -15)           # PCM16 grid
      tag = f"kind={kind} bs={bs} ch={ch} rate={rate} B={B} K={K} calls={calls} mode={mode} p0={p0:.2f} p1={p1:.2f} transient={transient} amp={amp} seed={seed}"
      slot = 2 * ch * bs + 16
      Kmax = K
      ks = [int(rng.integers(1, Kmax + 1)) for _ in range(calls)] if rng.random() < 0.5 else [K] * calls      # blocks per call may vary
      total = sum(ks)
      pcm = pcm[:, : total * bs]
      enc = ulc_amd.BatchEncoder(B, ch, bs, rate, Kmax)
      try:
          dec = ulc_amd.BatchDecoder(B, ch, bs, Kmax)
      except Exception:
          dec = None
      outs = []; k0 = 0
      for kc in ks:
          outs.append(enc.encode(pcm[:, k0 * bs:(k0 + kc) * bs], mode, p0, p1)); k0 += kc
      out = np.concatenate([o[0] for o in outs], axis=1); bits = np.concatenate([o[1] for o in outs], axis=1)
      wc = np.concatenate([o[2] for o in outs], axis=1); cplx = np.concatenate([o[3] for o in outs], axis=1)
      for s in range(B):
          ref = oracle_encode_debug(pcm[s], bs, rate, mode, p0, p1, slot=slot)
          assert np.array_equal(wc[s], ref["wc"]), f"{tag}: stream {s} WindowCtrl"
          assert cplx[s].tobytes() == ref["cplx"].tobytes(), f"{tag}: stream {s} BlockComplexity"
          assert np.array_equal(bits[s], ref["bits"]), f"{tag}: stream {s} sizes {bits[s]} vs {ref['bits']}"
          for k in range(total):
              nb = bits[s, k] // 8
              assert np.array_equal(out[s, k, :nb], ref["out"][k, :nb]), f"{tag}: stream {s} block {k} bytes"
      if dec is not None:
          got = []; k0 = 0
          for kc in ks:
              got.append(dec.decode(out[:, k0:k0 + kc])[0]); k0 += kc
          got = np.concatenate(got, axis=1)
          for s in range(B):
              rc, rp, rb = oracle_decode_stream(out[s], ch, bs)
              assert rc == 0 and np.array_equal(got[s].view(np.uint32), rp.view(np.uint32)), f"{tag}: stream {s} decoded PCM"
          dec.close()
      enc.close()
      n += 1; nblk += B * total
  msg = f"fuzz_parity: {n} random configurations, {nblk} blocks, all bit-exact (encode stream/WindowCtrl/complexity, decode PCM) in {time.time()-t0:.0f} s"
  print(msg)
  return n, nblk


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
