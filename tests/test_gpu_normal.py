"""The bootstrap multipliers made on the device (rocco_amd/csrc/normal.hip) against NumPy / SciPy themselves:
`Generator.standard_normal` over PCG64 value for value and draw for draw, and the smoothed, standardised multiplier rows
of rocco/inference.py:546-575 against the host restatement (rocco_amd.budget._generate_dependent_wild_weights, which is
pinned to the reference by tests/golden/budget_null_vectors.npz)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TAIL = 3.6541528853610088  # the ziggurat's base strip ends here: beyond it NumPy's values come through log1p


def _compare(seed, count):
    import torch

    from rocco_amd import budget

    rng_dev, rng_ref = np.random.default_rng(seed), np.random.default_rng(seed)
    got = budget.device_standard_normal(rng_dev, count).cpu().numpy()
    ref = rng_ref.standard_normal(count)
    assert got.shape == ref.shape
    tail = np.abs(ref) > TAIL
    # everything that does not go through log1p: NumPy's bits
    assert np.array_equal(got[~tail], ref[~tail])
    # the tail: the device's log1p against the host libm's -- at most the last place
    if tail.any():
        ulps = np.abs(got[tail].view(np.int64) - ref[tail].view(np.int64))
        assert ulps.max() <= 1, ulps.max()
    # the generator is where NumPy's is: the next host draws are the same
    assert np.array_equal(rng_dev.standard_normal(16), rng_ref.standard_normal(16))
    assert rng_dev.bit_generator.state == rng_ref.bit_generator.state
    return int(tail.sum()), int((got[tail] != ref[tail]).sum()) if tail.any() else 0


@pytest.mark.parametrize("count", [1, 2, 255, 256, 257, 511, 513, 4099, 100003])
def test_standard_normal_is_numpys_stream(gpu, count):
    for seed in (0, 1, 20240, 104729 * 3 + 11):
        _compare(seed, count)


def test_standard_normal_hundred_million_draws(gpu, monkeypatch):
    """10^8 values (about 26 000 of them through the tail loop, 1.5 million through a wedge): the values and the final
    generator state.  Round 5: the tail values are computed again on the host with the libm log1p this host's NumPy calls
    (rocco_amd/csrc/normal.hip): ZERO of the 10^8 differ from NumPy's.  ROCCO_HIP_NORMAL_TAIL=device keeps the device's
    log1p: then some of the tail values are one place off (130 in round 4's run), never more."""
    tails, differing = _compare(987654321, 100_000_000)
    assert tails > 20000
    assert differing == 0, f"{differing} of {tails} tail values differ from NumPy's"
    monkeypatch.setenv("ROCCO_HIP_NORMAL_TAIL", "device")
    tails_dev, differing_dev = _compare(987654321, 20_000_000)
    print(f"device log1p: tail values {tails_dev}, of which {differing_dev} differ from NumPy's in the last place")


def test_every_value_is_numpys_on_short_and_ragged_counts(gpu):
    """With the host's log1p behind the tail values, array_equal over everything -- counts around the chunk size, several
    seeds, 4 M values each (about a thousand tail values per run)."""
    from rocco_amd import budget

    for seed in (3, 77, 20240):
        for count in (4_000_001, 65_537):
            rng_dev, rng_ref = np.random.default_rng(seed), np.random.default_rng(seed)
            got = budget.device_standard_normal(rng_dev, count).cpu().numpy()
            assert np.array_equal(got, rng_ref.standard_normal(count)), (seed, count)
            assert rng_dev.bit_generator.state == rng_ref.bit_generator.state


def test_successive_calls_continue_the_stream(gpu):
    from rocco_amd import budget

    rng_dev, rng_ref = np.random.default_rng(5), np.random.default_rng(5)
    pieces = [budget.device_standard_normal(rng_dev, k).cpu().numpy() for k in (1000, 1, 77777, 300)]
    host_between = rng_dev.standard_normal(10)  # host draws in between continue the same stream
    pieces.append(host_between)
    pieces.append(budget.device_standard_normal(rng_dev, 5000).cpu().numpy())
    ref = rng_ref.standard_normal(sum(p.size for p in pieces))
    got = np.concatenate(pieces)
    assert np.array_equal(got, ref)  # (tail values included: the host's log1p, as NumPy)


@pytest.mark.parametrize("rows,n,hint", [(1, 5000, None), (3, 40000, None), (4, 1200, 101), (2, 300, 250), (5, 100001, 101),
                                          (2, 3000, 2500)])
def test_multipliers_match_the_host_rows(gpu, rows, n, hint):
    from rocco_amd import budget

    taps = budget._build_budget_bootstrap_kernel(budget._resolve_budget_bootstrap_bandwidth(n, hint))
    rng_dev, rng_ref = np.random.default_rng(31 + n), np.random.default_rng(31 + n)
    got = budget.device_multipliers(rng_dev, rows, n, taps)
    assert got is not None and tuple(got.shape) == (rows, n)
    got = got.cpu().numpy()
    ref = np.stack([budget._generate_dependent_wild_weights(n, taps, rng_ref) for _ in range(rows)])
    assert np.max(np.abs(got - ref)) <= 1e-12
    assert np.all(np.abs(got.mean(axis=1)) <= 1e-12) and np.all(np.abs(got.std(axis=1) - 1.0) <= 1e-12)
    assert rng_dev.bit_generator.state == rng_ref.bit_generator.state
