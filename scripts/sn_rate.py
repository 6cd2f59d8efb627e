"""The spectral-norm group step of the generator (18 convs) and of one texture discriminator: HIP-event time of the two entry
points (sn_power_iter = 3 kernels, weight_prep_batched = 1) and the bytes they have to move."""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("2dimageto3dmodel_amd"); gan = importlib.import_module("2dimageto3dmodel_amd.gan")
L = importlib.import_module("2dimageto3dmodel_amd._lib")
args = argparse.Namespace(norm_g="syncbatch", norm_d="none", conditional_class=True, conditional_color=False, conditional_text=False,
                          n_classes=[200], texture_resolution=256, mask_output=True, num_discriminators=2, texture_only=False, text_embedding_dim=256)
torch.manual_seed(3)
G = gan.Generator(args, 64, symmetric=True, mesh_head=True).cuda().train()
grp = G._sn_group()
wbytes = sum(c.weight_orig.numel() * 4 for c in grp.convs)
for name, fn in (("generator SN group (18 convs)", lambda: grp.step(True)),):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    L.enable_kernel_timers(True)
    for _ in range(50): fn()
    torch.cuda.synchronize()
    kt = L.collect_kernel_timers(); L.enable_kernel_timers(False)
    print(f"{name}: weights {wbytes / 1e6:.1f} MB fp32")
    for k, v in kt.items():
        print(f"   {k:24s} {v[1] / v[0] * 1e3:7.1f} us per call")
