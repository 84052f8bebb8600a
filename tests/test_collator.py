"""Host-side collator vs vectors produced by the reference's own collator (tools/make_golden.py). CPU only."""
import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd.collator import DataCollatorSpeechSeq2SeqWithPadding, make_prepare_dataset
from tests.util import golden

CASES = {
    "bos_all": [[50258, 1, 2, 3], [50258, 4, 5], [50258, 6]],
    "bos_some": [[50258, 1, 2, 3], [7, 4, 5], [50258, 6]],
    "single": [[50258, 9, 8, 7, 6, 5]],
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("as_list", [False, True])
def test_collator_matches_reference(name, as_list):
    G = golden("collator.npz")
    feats = []
    for i, lab in enumerate(CASES[name]):
        f = np.full((80, 3000), 0.25 * (i + 1), dtype=np.float32)
        f[i, i] = -1.0
        feats.append({"input_features": f.tolist() if as_list else f, "labels": lab})
    coll = DataCollatorSpeechSeq2SeqWithPadding(processor=None, decoder_start_token_id=50258)
    batch = coll(feats)
    assert batch["labels"].dtype == torch.int64
    np.testing.assert_array_equal(batch["labels"].numpy(), G[f"{name}/labels"])
    assert list(batch["input_features"].shape) == list(G[f"{name}/input_shape"])
    assert batch["input_features"].dtype == torch.float32
    np.testing.assert_array_equal(batch["input_features"][:, :4, :4].numpy(), G[f"{name}/input_probe"])


def test_collator_rejects_ragged_features():
    coll = DataCollatorSpeechSeq2SeqWithPadding(processor=None, decoder_start_token_id=50258)
    with pytest.raises(ValueError):
        coll([{"input_features": np.zeros((80, 3000), np.float32), "labels": [1]},
              {"input_features": np.zeros((80, 400), np.float32), "labels": [1]}])


def test_prepare_dataset_keeps_reference_shape():
    class P:
        def __call__(self, audio, sampling_rate, text):
            assert sampling_rate == 16000
            return {"input_features": [np.zeros((80, 3000), np.float32)], "labels": [50258, 1, 2]}

    out = make_prepare_dataset(P())({"audio": {"array": np.zeros(10), "sampling_rate": 16000}, "sentence": "x"})
    assert out["input_features"].shape == (80, 3000) and out["labels"] == [50258, 1, 2]
