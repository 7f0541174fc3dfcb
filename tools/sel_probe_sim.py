"""Offline sizing of k_select_wave's search (round 6): how many FULL probes (64 compares + 128 scalar instructions per wave each)
the bit-by-bit search needs on real keys - the oracle's final keys of six seeded stereo streams, 192 blocks, VBR 50 - against
bracketing T first from a sample of the keys.  CPU only (numpy + the oracle).  Printed on the round's data:
  bitwise (what shipped until round 5)            13.6 full probes
  interpolation alone, ordered-uint space         20.6   (log-domain keys: the count is nowhere near linear in the key)
  128 sample keys taken from registers 4g+1, 4g+3 by lane group g = lane/4, ONE bit descent on the sample (15.6 probes of 2 compares),
  its two values as the first full probes, interpolation (Illinois) after that:
      until <= 128 keys are left (a lane's own candidate list)   4.6 full probes
      until <= 512 (dense candidate list)                        2.2
The kernel (csrc/ulcx_enc_psy.hip, select_body) does the last line.  usage: python tools/sel_probe_sim.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from ulc_testlib import synth_pcm, oracle_encode_debug
def key_ord(f):
    b = f.view(np.uint32).astype(np.uint64)
    neg = (b >> 31) == 1
    return np.where(neg, (~b) & 0xFFFFFFFF, b | 0x80000000).astype(np.uint64)
blocks = []
for sid in range(6):
    pcm = synth_pcm(200 + sid, 32 * 2048, 2, 44100, transient=True, seed=3)
    r = oracle_encode_debug(pcm, 2048, 44100, 0, 50.0)
    for k in range(32):
        blocks.append((key_ord(r['keys'][k]), int(r['nout'][k])))
print(len(blocks), 'blocks; nout mean', np.mean([b[1] for b in blocks]))
SEL_CAND = 128
def sim_bitwise(u, k):
    mn, mx = int(u.min()), int(u.max())
    dif = mn ^ mx
    if dif == 0: return 0, 0
    bit = dif.bit_length() - 1
    T = mx & ~((2 << bit) - 1)
    cLo, cHi = len(u), 0
    full = 0; cand = 0; compacted = False
    while bit >= 0:
        t = T | (1 << bit)
        cnt = int((u >= t).sum())
        if compacted: cand += 1
        else: full += 1
        if cnt == k: break
        if cnt > k: T = t; cLo = cnt
        else: cHi = cnt if not compacted else cHi
        if not compacted and bit > 0 and cLo - cHi <= SEL_CAND: compacted = True
        bit -= 1
    return full, cand
def sim_interp(u, k, sample_first=False, illinois=True):
    """bracket [lo,hi) on arbitrary probe values until <= SEL_CAND keys inside; returns full probes"""
    uf = u
    lo, hi = int(u.min()), int(u.max()) + 1
    cLo, cHi = len(u), 0
    full = 0
    wLo = wHi = 1.0
    last = 0
    if sample_first:
        # diagonal sample of 64 keys, sorted; probe ranks around k*64/N
        idx = (np.arange(64) * 64 + np.arange(64)) % len(u)
        s = np.sort(u[idx])[::-1]
        q = k * 64.0 / len(u)
        for rank in (int(max(0, np.floor(q - 2.5))), int(min(63, np.ceil(q + 3.5)))):
            t = int(s[rank])
            if not (lo < t < hi): continue
            cnt = int((u >= t).sum()); full += 1
            if cnt == k: return full
            if cnt > k: lo, cLo = t, cnt
            else: hi, cHi = t, cnt
    while cLo - cHi > SEL_CAND and hi - lo > 1:
        # interpolation in ordered-uint space with Illinois weights; fall back to midpoint when the window did not halve
        a = (cLo - k) * wLo; b = (k - cHi) * wHi
        frac = a / (a + b) if (a + b) > 0 else 0.5
        t = lo + int((hi - lo) * frac)
        t = min(max(t, lo + 1), hi - 1)
        cnt = int((u >= t).sum()); full += 1
        if cnt == k: return full
        if cnt > k:
            lo, cLo = t, cnt
            if illinois: wHi = wHi * 0.5 if last == 1 else 1.0; wLo = 1.0
            last = 1
        else:
            hi, cHi = t, cnt
            if illinois: wLo = wLo * 0.5 if last == -1 else 1.0; wHi = 1.0
            last = -1
        if full > 40: break
    return full
res = {'bitwise': [], 'interp': [], 'interp_noill': [], 'sample+interp': []}
for u, k in blocks:
    if k <= 0: continue
    res['bitwise'].append(sim_bitwise(u, k)[0])
    res['interp'].append(sim_interp(u, k))
    res['interp_noill'].append(sim_interp(u, k, illinois=False))
    res['sample+interp'].append(sim_interp(u, k, sample_first=True))
for n, v in res.items():
    v = np.array(v); print('%-16s full probes mean %.2f  median %d  p90 %d  max %d' % (n, v.mean(), np.median(v), np.percentile(v, 90), v.max()))

print('--- variants')
def sim2(u, k, ns=64, dlo=2.5, dhi=3.5, mode='interp', cand=128):
    lo, hi = int(u.min()), int(u.max()) + 1
    cLo, cHi = len(u), 0
    full = 0
    N = len(u)
    step = N // ns
    idx = (np.arange(ns) * step + (np.arange(ns) * 37) % step) % N        # stratified: one key per stretch of N/ns, position varying
    s = np.sort(u[idx])[::-1]
    q = k * float(ns) / N
    for rank in (int(max(0, np.floor(q - dlo))), int(min(ns - 1, np.ceil(q + dhi)))):
        t = int(s[rank])
        if not (lo < t < hi): continue
        cnt = int((u >= t).sum()); full += 1
        if cnt == k: return full
        if cnt > k: lo, cLo = t, cnt
        else: hi, cHi = t, cnt
    wLo = wHi = 1.0; last = 0
    while cLo - cHi > cand and hi - lo > 1:
        if mode == 'interp':
            a = (cLo - k) * wLo; b = (k - cHi) * wHi
            frac = a / (a + b) if (a + b) > 0 else 0.5
        else: frac = 0.5
        t = lo + int((hi - lo) * frac); t = min(max(t, lo + 1), hi - 1)
        cnt = int((u >= t).sum()); full += 1
        if cnt == k: return full
        if cnt > k:
            lo, cLo = t, cnt; wHi = wHi * 0.5 if last == 1 else 1.0; wLo = 1.0; last = 1
        else:
            hi, cHi = t, cnt; wLo = wLo * 0.5 if last == -1 else 1.0; wHi = 1.0; last = -1
        if full > 40: break
    return full
for kw in (dict(ns=64), dict(ns=64, mode='mid'), dict(ns=128), dict(ns=128, dlo=3.5, dhi=4.5), dict(ns=256, dlo=5, dhi=6), dict(ns=64, cand=256), dict(ns=64, dlo=1.5, dhi=2.5), dict(ns=64, dlo=4, dhi=5)):
    v = np.array([sim2(u, k, **kw) for u, k in blocks if k > 0])
    print(kw, 'mean %.2f median %d p90 %d max %d' % (v.mean(), np.median(v), np.percentile(v, 90), v.max()))

print('--- register-pick samples (lane group g = lane >> 2 takes register 4g + off)')
def sim3(u, k, offs=(1,), dlo=2.5, dhi=3.5, cand=128):
    N = len(u)
    lane = np.arange(64)
    idx = np.concatenate([((4 * (lane >> 2) + o) * 64 + lane) for o in offs])
    ns = len(idx)
    lo, hi = int(u.min()), int(u.max()) + 1
    cLo, cHi = N, 0
    full = 0
    s = np.sort(u[idx])[::-1]
    q = k * float(ns) / N
    for rank in (int(max(0, np.floor(q - dlo))), int(min(ns - 1, np.ceil(q + dhi)))):
        t = int(s[rank])
        if not (lo < t < hi): continue
        cnt = int((u >= t).sum()); full += 1
        if cnt == k: return full
        if cnt > k: lo, cLo = t, cnt
        else: hi, cHi = t, cnt
    wLo = wHi = 1.0; last = 0
    while cLo - cHi > cand and hi - lo > 1:
        a = (cLo - k) * wLo; b = (k - cHi) * wHi
        frac = a / (a + b) if (a + b) > 0 else 0.5
        t = lo + int((hi - lo) * frac); t = min(max(t, lo + 1), hi - 1)
        cnt = int((u >= t).sum()); full += 1
        if cnt == k: return full
        if cnt > k:
            lo, cLo = t, cnt; wHi = wHi * 0.5 if last == 1 else 1.0; wLo = 1.0; last = 1
        else:
            hi, cHi = t, cnt; wLo = wLo * 0.5 if last == -1 else 1.0; wHi = 1.0; last = -1
        if full > 40: break
    return full
for kw in (dict(offs=(1,)), dict(offs=(1, 3)), dict(offs=(1, 3), dlo=3.5, dhi=4.5), dict(offs=(0, 1, 2, 3), dlo=5, dhi=6), dict(offs=(1,3), cand=256)):
    v = np.array([sim3(u, k, **kw) for u, k in blocks if k > 0])
    print(kw, 'mean %.2f median %d p90 %d max %d' % (v.mean(), np.median(v), np.percentile(v, 90), v.max()))

print('--- single descent on the sample (lower rank exact, upper bound from the descent path), reg-pick sample offs (1,3)')
def sim4(u, k, dlo=2.5, dhi=3.5, cand=128, both=False):
    N = len(u); lane = np.arange(64)
    idx = np.concatenate([((4 * (lane >> 2) + o) * 64 + lane) for o in (1, 3)])
    smp = u[idx]; ns = 128
    mn, mx = int(u.min()), int(u.max())
    if mn == mx: return 0, 0
    q = k * float(ns) / N
    kkLo = int(min(ns - 1, np.ceil(q + dhi))) + 1          # count wanted at the lower bracket value
    kkHi = int(max(0, np.floor(q - dlo))) + 1
    bit = (mn ^ mx).bit_length() - 1
    T = mx & ~((2 << bit) - 1)
    sprobes = 0; tHi = None
    for b in range(bit, -1, -1):
        t = T | (1 << b); c = int((smp >= t).sum()); sprobes += 1
        if c >= kkLo: T = t
        if c < kkHi and (tHi is None or t < tHi): tHi = t
        if c == kkLo: break
    tLo = T
    if both:
        T2 = mx & ~((2 << bit) - 1)
        for b in range(bit, -1, -1):
            t = T2 | (1 << b); c = int((smp >= t).sum()); sprobes += 1
            if c >= kkHi: T2 = t
            if c == kkHi: break
        tHi = T2 if q - dlo >= 0 else None
    lo, hi = mn, mx + 1; cLo, cHi = N, 0; full = 0
    for t in (tLo, tHi):
        if t is None or not (lo < t < hi): continue
        cnt = int((u >= t).sum()); full += 1
        if cnt == k: return full, sprobes
        if cnt > k: lo, cLo = t, cnt
        else: hi, cHi = t, cnt
    wLo = wHi = 1.0; last = 0
    while cLo - cHi > cand and hi - lo > 1:
        a = (cLo - k) * wLo; b = (k - cHi) * wHi
        frac = a / (a + b) if (a + b) > 0 else 0.5
        t = lo + int((hi - lo) * frac); t = min(max(t, lo + 1), hi - 1)
        cnt = int((u >= t).sum()); full += 1
        if cnt == k: return full, sprobes
        if cnt > k:
            lo, cLo = t, cnt; wHi = wHi * 0.5 if last == 1 else 1.0; wLo = 1.0; last = 1
        else:
            hi, cHi = t, cnt; wLo = wLo * 0.5 if last == -1 else 1.0; wHi = 1.0; last = -1
        if full > 40: break
    return full, sprobes
for kw in (dict(), dict(both=True), dict(dlo=1.5, dhi=2.5), dict(both=True, dlo=1.5, dhi=2.5), dict(both=True, dlo=3.5, dhi=4.5)):
    v = np.array([sim4(u, k, **kw) for u, k in blocks if k > 0])
    print(kw, 'full mean %.2f p90 %d max %d | sample probes mean %.1f' % (v[:,0].mean(), np.percentile(v[:,0], 90), v[:,0].max(), v[:,1].mean()))

print('--- dense compaction: candidate capacity 512')
for kw in (dict(cand=256), dict(cand=384), dict(cand=512), dict(cand=512, dlo=3.5, dhi=4.5), dict(cand=512, dlo=1.5, dhi=2.5)):
    v = np.array([sim4(u, k, **kw) for u, k in blocks if k > 0])
    print(kw, 'full mean %.2f p90 %d max %d | sample probes mean %.1f' % (v[:,0].mean(), np.percentile(v[:,0], 90), v[:,0].max(), v[:,1].mean()))
