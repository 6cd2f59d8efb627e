"""diagnostic: the G step of the SECOND cycle, second stream on vs off (deterministic mode): which intermediate first differs"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
pkg = importlib.import_module("2dimageto3dmodel_amd")
train = importlib.import_module("2dimageto3dmodel_amd.train")
gops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
CV = importlib.import_module("2dimageto3dmodel_amd.conv")
import test_gan_modules as T

batches = T._cycle_batches(4, 128, seed0=6100)
pkg.set_deterministic(True)
PRE = int(os.environ.get("DIAG_PRE", "3"))


def run(streams):
    gops.STREAMS_ON = streams
    torch.manual_seed(616)
    tr = train.GanTrainer(T._trainer_args(), device="cuda:0", mesh_template=None)
    tr.train()
    for i in range(PRE):
        b, z = batches[i % 3]
        tr.iteration(*b, noise=z, epoch=0)
    tr.finish_pending()
    out = {}
    (X_tex, X_alpha, X_mesh, C), z = batches[0]
    d_params = [p for p in tr.discriminator.parameters()]
    for p in d_params:
        p.requires_grad_(False)
    tr.optimizer_g.zero_grad(set_to_none=True)
    pred_tex, pred_mesh = tr.generator(z, C, None)
    pred_tex.retain_grad(); pred_mesh.retain_grad()
    out["fwd.pred_tex"], out["fwd.pred_mesh"] = pred_tex.detach().clone(), pred_mesh.detach().clone()
    disc, mask = tr.discriminator(gops.MaskedInput(pred_tex, X_alpha), pred_mesh, C, None)
    for k, dsc in enumerate(disc):
        for j, t in enumerate(dsc if isinstance(dsc, (list, tuple)) else [dsc]):
            out[f"fwd.disc{k}.{j}"] = t.detach().clone()
    loss = tr.criterion_gan(disc, True, for_discriminator=False, mask=mask, weight=tr._d_weight()).mean()
    out["fwd.loss"] = loss.detach().clone()
    with CV.deferred_wgrad_finish():
        loss.backward()
    out["bwd.dpred_tex"], out["bwd.dpred_mesh"] = pred_tex.grad.clone(), pred_mesh.grad.clone()
    for k, p in tr.generator.named_parameters():
        if p.grad is not None:
            out["grad." + k] = p.grad.detach().clone()
    torch.cuda.synchronize()
    return out


a, a2, b = run(True), run(True), run(False)
for name, (x, y) in (("on vs on ", (a, a2)), ("on vs off", (a, b))):
    bad = [k for k in x if not torch.equal(x[k], y[k])]
    print(name, "differing:", len(bad), "of", len(x))
    for k in bad[:60]:
        print("    ", k, tuple(x[k].shape), f"{(x[k].float() - y[k].float()).abs().max().item():.3e}")
