"""Text data format of the reference (chunk files) and its header quirks -- CPU only."""
import os

import numpy as np
import pytest

from cugp_amd import dataset


def test_roundtrip_and_cache(tmp_path):
    rng = np.random.default_rng(0)
    X, y = rng.uniform(-10, 10, (37, 10)), rng.standard_normal(37)
    a, b = str(tmp_path / "c0.txt"), str(tmp_path / "l0.txt")
    dataset.write_chunk(a, b, X, y)
    X2, y2 = dataset.load_chunk(a, b)
    assert X2.shape == (37, 10) and np.allclose(X2, X, rtol=1e-4) and np.allclose(y2, y, rtol=1e-4, atol=1e-6)
    assert os.path.exists(a + ".npz")
    X3, y3 = dataset.load_chunk(a, b, rows=20)                 # served from the cache
    assert np.array_equal(X3, X2[:20]) and np.array_equal(y3, y2[:20])
    with pytest.raises(ValueError):
        dataset.load_chunk(a, b, rows=38)


@pytest.mark.parametrize("header", ["256 10", "6000 1", "12.0 10", "4 10"])
def test_header_is_not_trusted(tmp_path, header):
    """sine_dataset_256_10 (hdr 256, 2000 rows), si6000_chunk1 (hdr '6000 1'), 1.py's float count."""
    X = np.arange(120, dtype=float).reshape(12, 10)
    a, b = str(tmp_path / "x.txt"), str(tmp_path / "y.txt")
    dataset.write_chunk(a, b, X, np.arange(12.0), header=header)
    X2, y2 = dataset.load_chunk(a, b)
    assert X2.shape == (12, 10) and np.array_equal(X2, X) and np.array_equal(y2, np.arange(12.0))


def test_shards_and_ownership(tmp_path):
    rng = np.random.default_rng(1)
    X, y = rng.uniform(-10, 10, (50, 3)), rng.standard_normal(50)
    parts = dataset.shard(X, y, 4)
    assert [p[0].shape[0] for p in parts] == [12, 12, 12, 12]          # remainder dropped, as 1.py does
    for i, (xs, ys) in enumerate(parts):
        dataset.write_chunk(str(tmp_path / ("in%d.txt" % i)), str(tmp_path / ("lab%d.txt" % i)), xs, ys)
    got = dataset.load_shards(str(tmp_path / "in"), str(tmp_path / "lab"), 4, only={1, 3})
    assert got[0] is None and got[2] is None
    assert np.allclose(got[1][0], parts[1][0], rtol=1e-4) and np.allclose(got[3][1], parts[3][1], rtol=1e-4, atol=1e-6)
