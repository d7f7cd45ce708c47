"""CPU checks of the lip-CNN restatement (oracle.cnn_forward, avsr/video.py:143-195) and of the product's op list.
The TF 'SAME' geometry is the part most easily got wrong: the oracle's convolutions are compared with a direct
NHWC loop implementation of the documented TF rule (out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); the odd
pixel goes to the bottom / right)."""
import numpy as np
import torch

from oracle import avsr_oracle as O


def _conv_same_ref(x, w, b, s):
    N, H, W, C = x.shape
    kh, kw, _, Co = w.shape
    Ho, Wo = -(-H // s), -(-W // s)
    pt = max((Ho - 1) * s + kh - H, 0) // 2
    pl = max((Wo - 1) * s + kw - W, 0) // 2
    y = np.zeros((N, Ho, Wo, Co))
    for ho in range(Ho):
        for wo in range(Wo):
            for i in range(kh):
                for j in range(kw):
                    h, ww = ho * s - pt + i, wo * s - pl + j
                    if 0 <= h < H and 0 <= ww < W:
                        y[:, ho, wo, :] += x[:, h, ww, :] @ w[i, j]
    return y + b


def test_same_padding_matches_tf_rule():
    rng = np.random.default_rng(0)
    for (H, k, s) in [(36, 3, 1), (36, 3, 2), (18, 3, 2), (9, 3, 2), (36, 1, 2), (9, 1, 2)]:
        x = rng.standard_normal((2, H, H, 3))
        w = rng.standard_normal((k, k, 3, 4))
        b = rng.standard_normal(4)
        pt, pb = O._same_pad(H, k, s)
        xt = torch.tensor(x).permute(0, 3, 1, 2)
        y = torch.nn.functional.conv2d(torch.nn.functional.pad(xt, (pt, pb, pt, pb)), torch.tensor(w).permute(3, 2, 0, 1),
                                       torch.tensor(b), stride=s).permute(0, 2, 3, 1).numpy()
        ref = _conv_same_ref(x, w, b, s)
        assert y.shape == ref.shape and np.abs(y - ref).max() < 1e-10, (H, k, s)


def test_cnn_shapes_and_parameter_names_agree_with_the_product():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("cnn_layout_only", os.path.join(os.path.dirname(__file__), "..", "avsr-tf1_amd", "cnn.py"),
                                                  submodule_search_locations=None)
    src = open(spec.origin).read()
    ns = {}
    # layout()/param_shapes() are pure Python: evaluate them without importing the GPU package
    exec(compile(src.split("class LipCNN")[0].replace("from . import ops", ""), spec.origin, "exec"), ns)
    cfg = O.OracleConfig(architecture="unimodal", video_units=(32,), audio_units=None, decoder_units=(32,), video_processing="resnet_cnn")
    P = O.init_params(cfg)
    names = {k[len("video/cnn/"):]: v.shape for k, v in P.items() if k.startswith("video/cnn/")}
    prod = {n: tuple(s) for n, s, _ in ns["param_shapes"](cfg.video_hw, cfg.cnn_filters, cfg.cnn_dense_units)}
    assert names == prod
    # 36 -> 36 -> 18 -> 9 -> 5, flatten kernel 5x5x64 -> 128
    assert P["video/cnn/flatten/kernel"].shape == (5, 5, 64, 128)
    feats = O.cnn_forward(O.to_torch(P), cfg, torch.zeros(3, 36, 36, 3, dtype=torch.float64), False, None)
    assert feats.shape == (3, 128)


def test_cnn_train_step_runs_and_regularises_conv_kernels():
    cfg = O.OracleConfig(architecture="unimodal", video_units=(16,), audio_units=None, decoder_units=(16,), embedding_size=8,
                         video_processing="resnet_cnn", cnn_filters=(4, 8), cnn_dense_units=8, video_feat=8, video_hw=(12, 12, 3))
    P = O.init_params(cfg)
    b = O.synthetic_batch(cfg, B=2, T_v=3, L=4, ragged=True)
    r = O.train_step(P, None, cfg, b)
    assert np.isfinite(r["loss"]) and np.abs(r["grads"]["video/cnn/layer0/kernel"]).max() > 0
    cfg0 = O.OracleConfig(**{**cfg.__dict__})
    Pz = {k: (np.zeros_like(v) if k.startswith("video/cnn/") and k.endswith("/kernel") else v) for k, v in P.items()}
    l2 = sum(0.5e-3 * float((P[k].astype(np.float64) ** 2).sum()) for k in O.cnn_l2_names(P))
    assert l2 > 0
    # moving statistics of the CNN's batch norms are updated with momentum 0.98
    assert not np.allclose(r["params"]["video/cnn/layer0_bn/moving_mean"], P["video/cnn/layer0_bn/moving_mean"])
