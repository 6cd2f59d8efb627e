"""the discriminator's first layers at N = 256 (a D step at batch 128 per GPU): tensors of 2 GiB -- 32-bit offsets end at 4 GiB,
signed ones at 2 GiB.  Each layer either passes the sampled parity check of tests/test_headline_batch_gpu.py or must refuse loudly."""
import importlib, os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
conv = importlib.import_module("2dimageto3dmodel_amd.conv")
import test_headline_batch_gpu as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prev_y = prev_b = None
for name, args, kw in (("D.conv1", (N, 256, 256, 8, 64, 5, 1, 2, 2, 2, 0, 1, T.LRELU), dict(want_dgrad=False)),
                       ("D.conv2", (N, 256, 256, 64, 128, 4, 2, 1, 1, 2, 0, 2, T.LRELU), dict()),
                       ("D.conv3", (N, 128, 128, 128, 256, 4, 2, 1, 1, 2, 0, 3, T.LRELU), dict()),
                       ("G.blk6.conv2", (N // 2, 256, 128, 64, 64, 3, 1, 1, 1, 1, 0, 15, 1.0), dict()),
                       ("G.blk6.conv1", (N // 2, 128, 64, 128, 64, 3, 1, 1, 1, 1, 1, 14, 1.0), dict())):
    try:
        if name in ("D.conv2", "D.conv3") and prev_y is not None:
            kw = dict(kw, bits_in=prev_b, x=prev_y)
        out = T._layer(conv, *args, **kw)
        if name.startswith("D."):
            prev_y, prev_b = out
        print(f"{name} N={args[0]}: OK ({out[0].numel() * 2 / 2**30:.2f} GiB output)")
    except Exception as e:  # noqa: BLE001
        prev_y = prev_b = None
        print(f"{name} N={args[0]}: {type(e).__name__}: {str(e)[:300]}")
    torch.cuda.empty_cache()
