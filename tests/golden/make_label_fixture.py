"""Writes tests/golden/reference_labels.json from the two label files the reference HOLDS
(/root/reference/data/train-labels-idx1-ubyte, t10k-labels-idx1-ubyte) -- with logs/trainLog.csv the only
reference-held data on this path (the image files are absent from the reference).  MNISTTrainer reads them at
MT:28-31 (open), MT:38-40 / MT:49-52 (magic 2049, count), MT:112-118 (one byte per label -> one-hot row).

Run in the BUILD CONTAINER only (the reference does not travel to the GPU box); the JSON is data -- header
integers, counts, a class histogram, the first 2 048 labels of each file, a checksum of the whole payload, and the
labels at the rows the trainer's sampler draws first (NNT:143-168 with Random(1), NNT:42) -- not source text.

    python -m tests.golden.make_label_fixture        # rewrites reference_labels.json

The parsing below is this script's own (struct + numpy): it must not depend on the code under test.  The sampler
draws come from the ORACLE's restatement (oracle/mlp_oracle.c), which tests/test_oracle.py pins to java.util.Random's
documented LCG.
"""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REF_DATA = "/root/reference/data"
FILES = {"train": "train-labels-idx1-ubyte", "t10k": "t10k-labels-idx1-ubyte"}
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_labels.json")
N_FIRST = 2048
N_DRAWS = 256      # two batches of 128 (BASELINE configs[1]) / eight of 32 (configs[0])


def fnv1a64(data):
    """FNV-1a, 64 bit (the checksum csrc/checkpoint.hip uses for its own files)."""
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def describe(path):
    raw = open(path, "rb").read()
    magic, n = struct.unpack(">ii", raw[:8])          # MT:76-80: four bytes, big endian
    payload = raw[8:]
    lab = np.frombuffer(payload, dtype=np.uint8)
    return {"file": os.path.basename(path), "file_bytes": len(raw), "magic": magic, "n": n,
            "payload_fnv1a64": "%016x" % fnv1a64(payload),
            "histogram": np.bincount(lab, minlength=10).tolist(),
            "min": int(lab.min()), "max": int(lab.max()),
            "first_labels": lab[:N_FIRST].tolist(), "last_labels": lab[-16:].tolist()}, lab


def build():
    from oracle import oracle
    oracle.build()
    out = {"source": "label files held by the reference under data/ (read by MNISTTrainer.java:28-31,38-40,49-52,112-118)",
           "generated_by": "tests/golden/make_label_fixture.py"}
    for key, name in FILES.items():
        d, lab = describe(os.path.join(REF_DATA, name))
        # the rows NeuralNetTrainer.sample draws first over this file's master list (row order), and their labels
        smp = oracle.Sampler(d["n"], seed=1)
        rows = np.concatenate([smp.sample(128) for _ in range(N_DRAWS // 128)])
        d["first_sampler_rows"] = rows.tolist()
        d["labels_at_first_sampler_rows"] = lab[rows].tolist()
        out[key] = d
    return out


if __name__ == "__main__":
    fx = build()
    with open(OUT, "w") as f:
        json.dump(fx, f, separators=(",", ":"))
        f.write("\n")
    print("wrote %s (%d bytes)" % (OUT, os.path.getsize(OUT)))
    for k in FILES:
        print(k, fx[k]["magic"], fx[k]["n"], fx[k]["histogram"], fx[k]["payload_fnv1a64"], fx[k]["first_sampler_rows"][:5])
