"""Real handwritten-digit data without network access: scikit-learn's bundled UCI optdigits
(1797 8x8 images, values 0..16) upsampled to 28x28 and written as MNIST idx files, so the whole
real-data path (idx reader -> upload -> training -> test-set evaluation) can be exercised.
This is NOT MNIST: accuracy on it says the pipeline learns and generalises on real images, it does
not pin the 99.2 % MNIST figure.   usage: python tools/digits_to_idx.py <out_dir> [n_test]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sklearn.datasets import load_digits
from distributed_tensorflow_amd.data import mnist

out = sys.argv[1]
n_test = int(sys.argv[2]) if len(sys.argv) > 2 else 297
os.makedirs(out, exist_ok=True)
d = load_digits()
img = np.kron(d.images, np.ones((3, 3)))                      # 24x24
img = np.pad(img, ((0, 0), (2, 2), (2, 2)))                   # 28x28
img = np.clip(img * (255.0 / 16.0), 0, 255).astype(np.uint8)
lab = d.target.astype(np.uint8)
rng = np.random.RandomState(0)
perm = rng.permutation(len(lab))
img, lab = img[perm], lab[perm]
n_train = len(lab) - n_test
mnist.write_idx_images(os.path.join(out, mnist.TRAIN_IMAGES), img[:n_train], gz=False)
mnist.write_idx_labels(os.path.join(out, mnist.TRAIN_LABELS), lab[:n_train], gz=False)
mnist.write_idx_images(os.path.join(out, mnist.TEST_IMAGES), img[n_train:], gz=False)
mnist.write_idx_labels(os.path.join(out, mnist.TEST_LABELS), lab[n_train:], gz=False)
print(f"wrote {n_train} train / {n_test} test images to {out}")
