"""HIP engine vs the COMMITTED fixtures (tests/golden/oracle_restatement_*.npz): the oracle does not run here,
so this test also works where /root/reference and a fast CPU are absent.  Fixtures come from the CPU restatement
("TF parity unpinned")."""
import dataclasses
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "oracle_restatement_*.npz"))))
def test_engine_matches_fixture(path):
    from avsr_tf1_amd.config import ModelConfig
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    z = np.load(path)
    kw = {k: (tuple(tuple(x) if isinstance(x, list) else x for x in v) if isinstance(v, list) else v)
          for k, v in json.loads(str(z["cfg_json"])).items()}
    names = {f.name for f in dataclasses.fields(ModelConfig)}
    cfg = ModelConfig(**{k: v for k, v in kw.items() if k in names})
    W = {k[2:]: z[k] for k in z.files if k.startswith("w:")}

    class NB:
        pass
    nb = NB()
    for k in z.files:
        if k.startswith("in:"):
            setattr(nb, k[3:], z[k])
    batch = Batch.from_numpy(nb)
    model = Seq2SeqModel(cfg, weights=W)
    logits = model.forward_train(batch)
    model.backward()
    model.apply_update()
    torch.cuda.synchronize()
    assert np.abs(logits.cpu().numpy() - z["out:logits"]).max() < 1e-4
    assert abs(float(model.loss.item()) - float(z["out:loss"])) < 1e-4
    assert abs(float(model.gnorm.item()) - float(z["out:global_norm"])) < 1e-4
    g = model.export_tf_weights("grads")
    for k in ("dec/out/kernel", "dec/l0/kernel"):
        ref = z["out:grad:" + k]
        assert np.abs(g[k] - ref).max() < 2e-4 * max(1e-3, np.abs(ref).max()) + 1e-6, k
    ids = Seq2SeqModel(cfg, weights=W).greedy_decode(batch, max_steps=8).cpu().numpy()
    assert ids.shape == z["out:greedy_ids"].shape and (ids == z["out:greedy_ids"]).all()
