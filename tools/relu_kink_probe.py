"""Round 6: seed 255 of the lip-CNN geometry fuzz (tests/test_gpu_fuzz.py) differed from the oracle in every CNN gradient by a few percent.
This probe walks the engine's own buffers: the input batch norm's backward, the flatten layer's relu mask / weight / bias gradient all agree with torch
recomputations from those buffers; the difference to the oracle is ONE ReLU input of +5.3e-7 (fp64) that the fp32 engine computes as <= 0.
python tools/relu_kink_probe.py [seed]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
torch.set_num_threads(8)
import test_gpu_fuzz as F
from test_gpu_model import make
from avsr_tf1_amd.model import Batch, Seq2SeqModel
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 255
rng = np.random.default_rng(9000 + seed)
hw = F.CNN_HW[int(rng.integers(len(F.CNN_HW)))]
filters = F.CNN_FILTERS[int(rng.integers(len(F.CNN_FILTERS)))]
while min(hw[0], hw[1]) < 2 ** (len(filters) - 1):
    filters = filters[:-1]
dense = int(rng.choice([8, 16, 32]))
case = "c4_bimodal_cnn" if rng.integers(2) else "c3_video_cnn_bi"
B, Tv = int(rng.integers(2, 5)), int(rng.integers(3, 7))
drop = bool(rng.integers(2))
print(seed, case, hw, filters, dense, B, Tv, drop)
for env in ({}, {"AVSR_CNN_FOLD": "0"}, {"AVSR_WG_WPC": "2"}):
    os.environ.update(env)
    O, ocfg, mcfg, W, batch = make(case, B=B, Ta=4 * Tv, Tv=Tv, L=5, video_hw=hw, cnn_filters=filters, cnn_dense_units=dense, video_feat=dense, use_dropout=drop)
    ref = O.train_step(W, None, ocfg, batch)
    model = Seq2SeqModel(mcfg, weights=W)
    db = Batch.from_numpy(batch)
    logits = model.forward_train(db); model.backward(); model.apply_update(); torch.cuda.synchronize()
    print(env, "gnorm", float(model.gnorm.item()), ref["global_norm"])
    grads = model.export_tf_weights("grads")
    for k, g in ref["grads"].items():
        scale = max(1e-3, np.abs(g).max()); err = np.abs(grads[k] - g).max()
        if err > 2e-4 * scale + 1e-6:
            print("   BAD", k, g.shape, "err %.3g scale %.3g" % (err, scale))
    cnn = model._cur[0]["enc"]["video"]["cnn"]
    print("   mfma", list(cnn.mfma), "direct", list(cnn.direct), "col", list(cnn.col), "fold", cnn.fold_wg, "bnb", cnn.bnb)
    for k in env: os.environ.pop(k)

# ---- where does it go wrong?  recompute the input batch norm's backward of the video stream in torch from the engine's own buffers
os.environ.pop("AVSR_CNN_FOLD", None)
O, ocfg, mcfg, W, batch = make(case, B=B, Ta=4 * Tv, Tv=Tv, L=5, video_hw=hw, cnn_filters=filters, cnn_dense_units=dense, video_feat=dense, use_dropout=drop)
model = Seq2SeqModel(mcfg, weights=W)
db = Batch.from_numpy(batch)
model.forward_train(db); model.backward(); torch.cuda.synchronize()
E = model._cur[0]["enc"]["video"]
x, dxn, dfeat = E["x"].double().cpu(), E["dxn"].double().cpu(), E["dfeat"].double().cpu()
Fd = dense
x2, dy = x.reshape(-1, Fd), dxn.reshape(-1, Fd)
mean, var = x2.mean(0), x2.var(0, unbiased=False)
inv = 1.0 / torch.sqrt(var + 1e-3)
gam = torch.tensor(W["video/bn/gamma"]).double()
xh = (x2 - mean) * inv
dx = gam * inv * (dy - dy.mean(0) - xh * (dy * xh).mean(0))
print("rows", x2.shape, "dfeat err vs torch BN backward:", float((dx - dfeat.reshape(-1, Fd)).abs().max()), "scale", float(dx.abs().max()))
print("engine mean err", float((E["mean"].double().cpu() - mean).abs().max()), "invstd err", float((E["invstd"].double().cpu() - inv).abs().max()))
cnn = E["cnn"]
gout = cnn.gmaps["out"].double().cpu().reshape(-1, Fd)
print("gmaps[out] vs dfeat", float((gout - dfeat.reshape(-1, Fd)).abs().max()))
out = cnn.maps["out"].double().cpu().reshape(-1, Fd)
dpre = cnn.pre_act.double().cpu().reshape(-1, Fd)
exp_dpre = gout * (out > 0)
print("dpre err", float((dpre - exp_dpre).abs().max()), "scale", float(exp_dpre.abs().max()))
src = [op for op in cnn.ops if op[0] == "flatten"][0][2]
xs = cnn.maps[src].double().cpu().reshape(out.shape[0], -1)
exp_gw = xs.t() @ exp_dpre
g = model.export_tf_weights("grads")
gw = torch.tensor(g["video/cnn/flatten/kernel"]).double().reshape(-1, Fd)
l2 = 0.0
print("flatten dW err (engine vs torch from engine buffers, no L2):", float((gw - exp_gw).abs().max()), "scale", float(exp_gw.abs().max()), "K", xs.shape)
gb = torch.tensor(g["video/cnn/flatten/bias"]).double()
print("flatten db err:", float((gb - exp_dpre.sum(0)).abs().max()), "scale", float(exp_dpre.sum(0).abs().max()))
r = O.train_step(W, None, ocfg, batch)
print("oracle flatten db vs torch-from-engine-buffers:", float((torch.tensor(r["grads"]["video/cnn/flatten/bias"]).double() - exp_dpre.sum(0)).abs().max()))
print("video_len", batch.video_len, "T_v", Tv)
Wf = torch.tensor(W["video/cnn/flatten/kernel"]).double().reshape(-1, Fd)
bf = torch.tensor(W["video/cnn/flatten/bias"]).double()
z64 = xs @ Wf + bf
flip = ((z64 > 0) != (out > 0))
print("flatten pre-activations: min |z| %.3g; relu masks that differ between fp64 z and the engine's fp32 output: %d" % (float(z64.abs().min()), int(flip.sum())))
idx = flip.nonzero()
for i_ in idx[:5]:
    print("   row %d feat %d z64 %.3e engine out %.3e  dfeat %.3e" % (int(i_[0]), int(i_[1]), float(z64[i_[0], i_[1]]), float(out[i_[0], i_[1]]), float(gout[i_[0], i_[1]])))
