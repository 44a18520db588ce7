"""Layout of the loss inputs as the reference's batch producer builds them (data_generator.py:391-426)."""
import numpy as np
import pytest

from rnnt_amd import batching


def test_pad_batch_layout():
    rng = np.random.default_rng(0)
    feats = [rng.standard_normal((t, 3)) for t in (5, 2, 7)]
    seqs = [[4, 1, 1], [], [0, 5, 2, 3]]
    inputs, outputs = batching.pad_batch(feats, seqs, vocab_size=6)
    assert set(inputs) == {"INPUT_TRANS", "INPUT_PREDS", "INPUT_LABEL", "INPUT_INLEN", "INPUT_LBLEN"}
    assert inputs["INPUT_TRANS"].shape == (3, 7, 3) and inputs["INPUT_PREDS"].shape == (3, 5, 6)
    assert inputs["INPUT_LABEL"].shape == (3, 4)
    assert inputs["INPUT_INLEN"].tolist() == [5, 2, 7] and inputs["INPUT_LBLEN"].tolist() == [3, 0, 4]
    assert inputs["INPUT_INLEN"].dtype == np.float64   # the model's int32 Inputs cast them (keras_model.py:125-127)
    np.testing.assert_array_equal(inputs["INPUT_TRANS"][1, :2], feats[1])
    assert not inputs["INPUT_TRANS"][1, 2:].any()
    assert inputs["INPUT_LABEL"][0].tolist() == [4, 1, 1, 0]          # padded with 0, not with the blank
    pred = inputs["INPUT_PREDS"]
    assert not pred[:, 0].any()                                       # the prepended blank is an all-zero row
    assert pred[0, 1:4].argmax(-1).tolist() == [4, 1, 1] and pred[0, 1:4].sum() == 3 and not pred[0, 4:].any()
    assert not pred[1].any()
    assert outputs["LOSS_RNNT"].tolist() == [0, 0, 0]


def test_pad_batch_rejects_malformed_input():
    f = [np.zeros((2, 3))]
    with pytest.raises(ValueError):
        batching.pad_batch(f, [[6]], vocab_size=6)
    with pytest.raises(ValueError):
        batching.pad_batch(f, [[1], [2]], vocab_size=6)
    with pytest.raises(ValueError):
        batching.pad_batch([np.zeros((0, 3))], [[1]], vocab_size=6)
    with pytest.raises(ValueError):
        batching.pad_batch([np.zeros((2, 3)), np.zeros((2, 4))], [[1], [1]], vocab_size=6)
