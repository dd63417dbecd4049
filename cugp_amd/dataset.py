"""Readers/writers for the reference's text data files and a binary cache.

Format (chunked_dataset/*.txt, scaling_dataset/*.txt of the reference): inputs file = one header line
"N D" then whitespace-separated rows; labels file = one value per line, no header.  The header is NOT
trusted (several shipped files disagree with theirs -- e.g. sine_dataset_256_10 says 256 and
holds 2000 rows, si6000_chunk1 says "6000 1", the sharder scaling_dataset/1.py writes a float count
"1500.0 10"); rows are counted from the data and D is taken from the first data line.  The reference
re-parses these text files on every evaluation (cuda_scalingdist/cg_solver.cpp:42-70); here a file is
parsed once and cached as .npz next to it (or under cache_dir).
"""
import os

import numpy as np


def read_inputs(path, dim=None):
    with open(path) as f:
        header = f.readline().split()
        body = f.read().split()
    vals = np.array(body, dtype=np.float64)
    if dim is None:
        # the header's D is usually right; verify against the first data line
        with open(path) as f:
            f.readline()
            first = f.readline().split()
        dim = len(first) if first else int(float(header[1]))
    if vals.size % dim:
        raise ValueError("%s: %d values are not a multiple of D=%d" % (path, vals.size, dim))
    return np.ascontiguousarray(vals.reshape(-1, dim))


def read_labels(path):
    return np.ascontiguousarray(np.loadtxt(path, dtype=np.float64).reshape(-1))


def load_chunk(inputs_path, labels_path, rows=None, cache_dir=None):
    """-> (X[n,d], y[n]); rows=None keeps everything the files hold (train + held-out test rows)."""
    key = os.path.basename(inputs_path) + ".npz"
    cpath = os.path.join(cache_dir or os.path.dirname(os.path.abspath(inputs_path)), key)
    src_m = max(os.path.getmtime(inputs_path), os.path.getmtime(labels_path))
    X = y = None
    if os.path.exists(cpath) and os.path.getmtime(cpath) >= src_m:
        try:
            z = np.load(cpath)
            X, y = z["X"], z["y"]
        except Exception:
            X = None
    if X is None:
        X, y = read_inputs(inputs_path), read_labels(labels_path)
        n = min(X.shape[0], y.shape[0])
        X, y = X[:n], y[:n]
        try:
            # several ranks may parse the same chunk at once (every rank loads chunk 0's held-out rows): write to
            # a private temporary and rename, so a reader never sees a half-written archive
            tmp = "%s.%d.tmp.npz" % (cpath, os.getpid())
            np.savez(tmp, X=X, y=y)
            os.replace(tmp, cpath)
        except OSError:
            pass                                   # read-only location: just skip the cache
    if rows is not None:
        if rows > X.shape[0]:
            raise ValueError("%s holds %d rows, %d requested" % (inputs_path, X.shape[0], rows))
        X, y = X[:rows], y[:rows]
    return np.ascontiguousarray(X), np.ascontiguousarray(y)


def load_shards(prefix_inputs, prefix_labels, numchunks, rows=None, cache_dir=None, only=None):
    """Chunk i lives in <prefix>i.txt (cuda_scalingdist/cg_solver.cpp:47-48).  `only`: iterable of chunk
    indices to actually read (the ones this rank owns); the others come back as None."""
    out = []
    for i in range(numchunks):
        if only is not None and i not in only:
            out.append(None)
            continue
        out.append(load_chunk("%s%d.txt" % (prefix_inputs, i), "%s%d.txt" % (prefix_labels, i), rows, cache_dir))
    return out


def write_chunk(inputs_path, labels_path, X, y, header=None):
    """Writes the reference's text format with 5 significant digits, as its generator did."""
    X = np.asarray(X, dtype=np.float64)
    with open(inputs_path, "w") as f:
        f.write((header or "%d %d" % X.shape) + "\n")
        for r in X:
            f.write(" ".join("%.5g" % v for v in r) + "\n")
    with open(labels_path, "w") as f:
        for v in np.asarray(y, dtype=np.float64):
            f.write("%.5g\n" % v)


def shard(X, y, numshards):
    """scaling_dataset/1.py, 2.py: contiguous shards of floor(N / numshards) rows (remainder dropped)."""
    n = X.shape[0] // numshards
    return [(np.ascontiguousarray(X[i * n:(i + 1) * n]), np.ascontiguousarray(y[i * n:(i + 1) * n]))
            for i in range(numshards)]
