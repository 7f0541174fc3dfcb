"""GPU debug: decode of a big batch with parts of the decoder switched off (ULCX_DBG_SKIP)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    import numpy as np, torch
    sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd")); sys.path.insert(0, ROOT)
    import ulc_amd, bench
    dev = torch.device("cuda", 0)
    B, K = int(sys.argv[1]), int(sys.argv[2])
    pcm = bench.make_pcm(torch, B, K * 2048, dev, seed=1)
    enc = ulc_amd.BatchEncoder(B, 2, 2048, 44100, K); dec = ulc_amd.BatchDecoder(B, 2, 2048, K)
    slot = enc.slot
    d_out = torch.zeros(B * K * slot, dtype=torch.uint8, device=dev); d_bits = torch.zeros(B * K, dtype=torch.int32, device=dev)
    d_dec = torch.zeros(B * K * 2048 * 2, dtype=torch.float32, device=dev); d_db = torch.zeros(B * K, dtype=torch.int32, device=dev)
    enc.encode_dev(pcm.data_ptr(), K, d_out.data_ptr(), d_bits.data_ptr(), p0=50.0); torch.cuda.synchronize()
    dec.decode_dev(d_out.data_ptr(), slot, K, d_dec.data_ptr(), d_db.data_ptr()); torch.cuda.synchronize()
    print("ok", B, K, os.environ.get("ULCX_DBG_SKIP"), int((d_db > 0).sum()), dec.stage_ms(), flush=True)
    sys.exit(0)
for B, K in [(4096, 16), (3072, 16), (4096, 12)]:
    for skip in ["8", "7", "6", "5", "3", "0"]:
        env = dict(os.environ, ULCX_DBG_SKIP=skip)
        r = subprocess.run([sys.executable, __file__, str(B), str(K)], env=env, capture_output=True, text=True)
        print(B, K, "skip", skip, "rc", r.returncode, r.stdout.strip()[-200:], r.stderr.strip()[-120:].replace("\n", " | "), flush=True)
