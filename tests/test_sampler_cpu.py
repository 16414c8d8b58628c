"""The epoch sampler of NeuralNetTrainer (NNT:143-168) as the library draws it -- a bit per row, 64 rows per word, a Fenwick
tree over the words, csrc/sampler.hip -- against the oracle's restatement (an array the drawn row is removed from), on the CPU:
the sampler is host code (no GPU call), and the same comparison runs in tests/test_trainer_gpu.py on the GPU box."""
import numpy as np
import pytest


@pytest.mark.parametrize("master,batch,draws", [(60000, 128, 40), (10, 4, 12), (37, 16, 20), (1000, 999, 5), (64, 63, 9), (65, 64, 9),
                                                (128, 17, 40), (129, 128, 7), (4097, 300, 60), (2, 1, 9)])
def test_library_sampler_equals_oracle_sampler(gnn, oracle_mod, master, batch, draws):
    a, b = gnn.Sampler(master), oracle_mod.Sampler(master)
    for _ in range(draws):
        x, y = a.sample(batch), b.sample(batch)
        assert np.array_equal(x, y)


def test_first_draw_known_answer(gnn):
    assert gnn.Sampler(60000).sample(5)[0] == 8985      # Random(1).nextInt(60000)


def test_epochs_are_permutations(gnn):
    """Within an epoch every row once: 6 batches of 50 over 300 rows are one permutation, then the next epoch starts."""
    s = gnn.Sampler(300, seed=7)
    for epoch in range(3):
        rows = np.concatenate([s.sample(50) for _ in range(6)])
        assert sorted(rows.tolist()) == list(range(300))
