"""The UrbanSound Transformer classifier's inference path on the native kernels (d 128, heads of 32, post-LN, S = T + 1)
against the torch.nn restatement of the reference module (oracle/urbansound_classifier.py)."""
import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd import weights as wts
from oracle.urbansound_classifier import ReferenceTransformerClassifier

pytestmark = pytest.mark.gpu


def _pair(n_mels, T, seed):
    from mlx8_ws_audio_transformer_amd.urbansound_classifier import TransformerUrbanSound8KClassifier
    torch.manual_seed(seed)
    ref = ReferenceTransformerClassifier(n_mels=n_mels).eval()
    with torch.no_grad():                       # make every parameter non-trivial (fresh LayerNorms / biases are 1 / 0)
        for i, (name, p) in enumerate(ref.named_parameters()):
            u = torch.from_numpy(wts.unit_variates("cls." + name, p.numel(), seed).reshape(p.shape).astype(np.float32))
            p.copy_(1.0 + 0.1 * u if "norm" in name and name.endswith("weight") else (p + 0.05 * u if p.dim() > 1 else 0.1 * u))
        ref.pos_embed = torch.nn.Parameter(0.02 * torch.randn(1, T + 1, 128)); ref.n_frames = T
    nat = TransformerUrbanSound8KClassifier(n_mels=n_mels)
    nat.pos_embed = torch.nn.Parameter(ref.pos_embed.detach().clone()); nat.n_frames = T
    missing, unexpected = nat.load_state_dict(ref.state_dict(), strict=True)      # same names as the reference module
    assert not missing and not unexpected
    return ref, nat.cuda().eval()


@pytest.mark.parametrize("n_mels,T,batch", [(128, 126, 3), (80, 501, 2), (64, 126, 5)])
def test_logits_and_features_match_reference_module(n_mels, T, batch):
    ref, nat = _pair(n_mels, T, seed=n_mels + T)
    x = torch.from_numpy(wts.unit_variates("cls.x", batch * n_mels * T, 3).reshape(batch, n_mels, T).astype(np.float32)) * 2.0 - 4.0
    with torch.no_grad():
        want_f, want = ref.features(x), ref(x)
    got_f, got = nat.get_feature_embeddings(x.cuda()).cpu(), nat(x.cuda()).cpu()
    assert tuple(got.shape) == (batch, 10) and tuple(got_f.shape) == (batch, 128)
    np.testing.assert_allclose(got_f.numpy(), want_f.numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-3)
    assert torch.equal(got.argmax(-1), want.argmax(-1))


def test_training_mode_raises_instead_of_falling_back():
    _, nat = _pair(128, 126, seed=1)
    nat.train()
    with pytest.raises(NotImplementedError):
        nat(torch.zeros(1, 128, 126).cuda())
