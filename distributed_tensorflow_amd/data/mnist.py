"""MNIST input pipeline with the semantics of TF's ``input_data`` helper
(reference .idea/MNISTDist.py:11, :167 ``read_data_sets(data_dir, one_hot=True)``
and :178 ``mnist.train.next_batch(batch_size)``).

* parses the four idx(.gz) files from ``data_dir`` (no download: this
  environment has no network; ``synthetic=True`` or a missing directory with
  ``allow_synthetic`` substitutes MNIST-shaped synthetic data),
* 55 000 / 5 000 / 10 000 train / validation / test split,
* images float32 in [0,1] flattened to 784, labels one-hot float32 or int,
* ``next_batch``: shuffle on first use, and at an epoch boundary return the
  rest of the old epoch followed by the head of a freshly shuffled one.

Every worker loads and shuffles the full set independently, exactly like the
reference (no sharding).
"""
from __future__ import annotations

import gzip
import os
import struct
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .synthetic import synthetic_mnist

TRAIN_IMAGES = "train-images-idx3-ubyte"
TRAIN_LABELS = "train-labels-idx1-ubyte"
TEST_IMAGES = "t10k-images-idx3-ubyte"
TEST_LABELS = "t10k-labels-idx1-ubyte"
VALIDATION_SIZE = 5000


def _open(path: str):
    if os.path.exists(path + ".gz"):
        return gzip.open(path + ".gz", "rb")
    if os.path.exists(path):
        return open(path, "rb")
    raise FileNotFoundError(path + "[.gz]")


def read_idx_images(path: str) -> np.ndarray:
    """idx3-ubyte -> uint8 [n, rows, cols]"""
    with _open(path) as f:
        magic, n, rows, cols = struct.unpack(">IIII", f.read(16))
        if magic != 2051:
            raise ValueError(f"bad magic {magic} in MNIST image file {path}")
        data = np.frombuffer(f.read(n * rows * cols), dtype=np.uint8)
    return data.reshape(n, rows, cols)


def read_idx_labels(path: str) -> np.ndarray:
    with _open(path) as f:
        magic, n = struct.unpack(">II", f.read(8))
        if magic != 2049:
            raise ValueError(f"bad magic {magic} in MNIST label file {path}")
        return np.frombuffer(f.read(n), dtype=np.uint8).copy()


def write_idx_images(path: str, images_u8: np.ndarray, gz: bool = True):
    n, r, c = images_u8.shape
    blob = struct.pack(">IIII", 2051, n, r, c) + images_u8.astype(np.uint8).tobytes()
    with (gzip.open(path + ".gz", "wb") if gz else open(path, "wb")) as f:
        f.write(blob)


def write_idx_labels(path: str, labels_u8: np.ndarray, gz: bool = True):
    blob = struct.pack(">II", 2049, labels_u8.shape[0]) + labels_u8.astype(np.uint8).tobytes()
    with (gzip.open(path + ".gz", "wb") if gz else open(path, "wb")) as f:
        f.write(blob)


def dense_to_one_hot(labels: np.ndarray, num_classes: int = 10) -> np.ndarray:
    out = np.zeros((labels.shape[0], num_classes), dtype=np.float32)
    out[np.arange(labels.shape[0]), labels.astype(np.int64)] = 1.0
    return out


class DataSet:
    """In-memory split with TF ``DataSet.next_batch`` semantics."""

    def __init__(self, images: np.ndarray, labels: np.ndarray, one_hot: bool = False, seed: Optional[int] = None):
        assert images.shape[0] == labels.shape[0]
        self._images = images.reshape(images.shape[0], -1).astype(np.float32)
        self._int_labels = labels.astype(np.int32) if labels.ndim == 1 else labels.argmax(-1).astype(np.int32)
        self._one_hot = one_hot
        self._labels = dense_to_one_hot(self._int_labels) if one_hot else self._int_labels
        self._num_examples = images.shape[0]
        self._epochs_completed = 0
        self._index_in_epoch = 0
        self._rng = np.random.RandomState(seed)

    images = property(lambda self: self._images)
    labels = property(lambda self: self._labels)
    int_labels = property(lambda self: self._int_labels)
    num_examples = property(lambda self: self._num_examples)
    epochs_completed = property(lambda self: self._epochs_completed)

    def _shuffle(self):
        perm = self._rng.permutation(self._num_examples)
        self._images = self._images[perm]
        self._labels = self._labels[perm]
        self._int_labels = self._int_labels[perm]

    def next_batch(self, batch_size: int, shuffle: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        start = self._index_in_epoch
        if self._epochs_completed == 0 and start == 0 and shuffle:
            self._shuffle()
        if start + batch_size > self._num_examples:
            self._epochs_completed += 1
            rest = self._num_examples - start
            img_rest, lab_rest = self._images[start:], self._labels[start:]
            if shuffle:
                self._shuffle()
            start = 0
            self._index_in_epoch = batch_size - rest
            end = self._index_in_epoch
            return (np.concatenate((img_rest, self._images[start:end]), axis=0),
                    np.concatenate((lab_rest, self._labels[start:end]), axis=0))
        self._index_in_epoch += batch_size
        end = self._index_in_epoch
        return self._images[start:end], self._labels[start:end]


@dataclass
class Datasets:
    train: DataSet
    validation: DataSet
    test: DataSet
    synthetic: bool = False


def mnist_files_present(data_dir: str) -> bool:
    def has(n):
        return os.path.exists(os.path.join(data_dir, n)) or os.path.exists(os.path.join(data_dir, n + ".gz"))
    return all(has(n) for n in (TRAIN_IMAGES, TRAIN_LABELS, TEST_IMAGES, TEST_LABELS))


def read_data_sets(data_dir: str, one_hot: bool = False, synthetic: bool = False, allow_synthetic: bool = True,
                   validation_size: int = VALIDATION_SIZE, seed: Optional[int] = None,
                   synthetic_sizes: Tuple[int, int] = (60000, 10000)) -> Datasets:
    if not synthetic and not mnist_files_present(data_dir):
        if not allow_synthetic:
            raise FileNotFoundError(
                f"MNIST idx files not found in {data_dir!r} and downloading is impossible here (no network); "
                "place train-images-idx3-ubyte(.gz) etc. there or pass --synthetic")
        synthetic = True
    if synthetic:
        n_train, n_test = synthetic_sizes
        xi, yi = synthetic_mnist(n_train + n_test, seed=1234)
        train_x, train_y, test_x, test_y = xi[:n_train], yi[:n_train], xi[n_train:], yi[n_train:]
    else:
        train_x = read_idx_images(os.path.join(data_dir, TRAIN_IMAGES)).astype(np.float32) / 255.0
        train_y = read_idx_labels(os.path.join(data_dir, TRAIN_LABELS))
        test_x = read_idx_images(os.path.join(data_dir, TEST_IMAGES)).astype(np.float32) / 255.0
        test_y = read_idx_labels(os.path.join(data_dir, TEST_LABELS))
    validation_size = min(validation_size, train_x.shape[0] // 2)
    val_x, val_y = train_x[:validation_size], train_y[:validation_size]
    train_x, train_y = train_x[validation_size:], train_y[validation_size:]
    return Datasets(train=DataSet(train_x, train_y, one_hot, seed), validation=DataSet(val_x, val_y, one_hot, seed),
                    test=DataSet(test_x, test_y, one_hot, seed), synthetic=synthetic)
