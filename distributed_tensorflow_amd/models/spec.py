"""Model specification: shapes, TF variable names and the flat parameter arena.

Shapes/layouts follow the reference graph exactly (reference
.idea/MNISTDist.py:117-141): conv filters HWIO, dense weights [in, out],
activations NHWC, flatten in HWC order.  The reference leaves every
tf.Variable unnamed, so its checkpoint keys are ``Variable``, ``Variable_1``...
in creation order (weights dict, biases dict, then global_step, M:117-147).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

# logical name -> TF checkpoint key (creation order in the reference script)
TF_VARIABLE_NAMES: Dict[str, str] = {
    "wc1": "Variable",
    "wc2": "Variable_1",
    "wd1": "Variable_2",
    "out_w": "Variable_3",
    "bc1": "Variable_4",
    "bc2": "Variable_5",
    "bd1": "Variable_6",
    "out_b": "Variable_7",
    "global_step": "Variable_8",
}

NUM_INPUT = 784    # reference M:33
NUM_CLASSES = 10   # reference M:34
IMAGE_SIZE = 28    # reference M:38


@dataclass(frozen=True)
class ModelSpec:
    """conv5x5-c1 -> pool -> conv5x5-c2 -> pool -> fc(hidden) -> dropout -> fc(classes)."""

    c1: int = 32
    c2: int = 64
    hidden: int = 1024
    classes: int = NUM_CLASSES
    name: str = "deep_mnist"

    @staticmethod
    def deep_mnist() -> "ModelSpec":
        return ModelSpec()

    @staticmethod
    def wide() -> "ModelSpec":
        """BASELINE.json config #5: widened CNN (conv 128/256, fc4096)."""
        return ModelSpec(c1=128, c2=256, hidden=4096, name="deep_mnist_wide")

    @staticmethod
    def from_name(name: str) -> "ModelSpec":
        if name in ("deep_mnist", "base", "mnist"):
            return ModelSpec.deep_mnist()
        if name in ("wide", "deep_mnist_wide"):
            return ModelSpec.wide()
        raise ValueError(f"unknown model {name!r}")

    @property
    def flat(self) -> int:
        return 7 * 7 * self.c2

    def shapes(self) -> Dict[str, Tuple[int, ...]]:
        return {
            "wc1": (5, 5, 1, self.c1),
            "wc2": (5, 5, self.c1, self.c2),
            "wd1": (self.flat, self.hidden),
            "out_w": (self.hidden, self.classes),
            "bc1": (self.c1,),
            "bc2": (self.c2,),
            "bd1": (self.hidden,),
            "out_b": (self.classes,),
        }

    def num_params(self) -> int:
        n = 0
        for s in self.shapes().values():
            k = 1
            for d in s:
                k *= d
            n += k
        return n

    def flops_per_image(self) -> Tuple[float, float]:
        """(forward, forward+backward) FLOPs per image, 2 per MAC."""
        conv1 = 2.0 * 28 * 28 * 25 * self.c1
        conv2 = 2.0 * 14 * 14 * 25 * self.c1 * self.c2
        fc1 = 2.0 * self.flat * self.hidden
        out = 2.0 * self.hidden * self.classes
        fwd = conv1 + conv2 + fc1 + out
        # backward: wgrad for all, dgrad for all but conv1 (its input is data)
        bwd = conv1 + 2 * (conv2 + fc1 + out)
        return fwd, fwd + bwd


def _align4(n: int) -> int:
    return (n + 3) & ~3


@dataclass
class ArenaLayout:
    """Flat fp32 arena holding every trainable tensor.

    ``wd1`` comes first: its gradient is 98 % of the payload and is final right after the
    fc1 wgrad GEMM, so ONE collective over [0, |wd1|) overlaps with the whole conv backward.
    Everything else (head = out_w|out_b|bd1, then conv2, conv1 -- 0.25 MB in total) follows
    contiguously and travels as one small tail collective, after one slab reduction and before one
    fused update.  Each tensor starts on a 16-byte boundary.
    The split-K slabs written by the wgrad kernels use the same relative layout
    as the arena segments they reduce into (head = out_w|out_b|bd1, conv2 =
    wc2|bc2, conv1 = wc1|bc1).

    The arena ends in a 4-float CONTROL block (``ctrl``): no tensor lives there.  In the gradient arena it
    carries the workers' stop requests inside the tail all-reduce of every synchronous step ("abort
    flag all-reduce"), so a stop is seen by every worker at the same step and costs no extra collective.
    """

    spec: ModelSpec
    order: Tuple[str, ...] = ("wd1", "out_w", "out_b", "bd1", "wc2", "bc2", "wc1", "bc1")
    offsets: Dict[str, int] = field(default_factory=dict)
    sizes: Dict[str, int] = field(default_factory=dict)
    total: int = 0
    ctrl: int = 0      # float offset of the control block (the last 4 floats of the arena)

    def __post_init__(self):
        shapes = self.spec.shapes()
        off = 0
        for name in self.order:
            n = 1
            for d in shapes[name]:
                n *= d
            self.offsets[name] = off
            self.sizes[name] = n
            off += _align4(n)
        self.ctrl = off
        self.total = off + 4

    def shape(self, name: str) -> Tuple[int, ...]:
        return self.spec.shapes()[name]

    def segment(self, first: str, last: str) -> Tuple[int, int]:
        """[start, end) float offsets covering tensors first..last (inclusive, with padding)."""
        start = self.offsets[first]
        end = self.offsets[last] + _align4(self.sizes[last])
        return start, end

    def view(self, arena, name: str):
        """A tensor view of ``name`` inside a flat torch arena."""
        o = self.offsets[name]
        return arena[o:o + self.sizes[name]].view(*self.shape(name))

    def names(self) -> List[str]:
        return list(self.order)
