"""distributed_tensorflow_amd: an MI355X-native distributed deep-MNIST trainer.

Capabilities mirror ellie-ba/Distributed_TensorFlow (reference
.idea/MNISTDist.py): the deep-MNIST CNN, the ps/worker flag surface, chief
managed tf.train.Saver-compatible checkpoints, async parameter-server and
synchronous data-parallel training.  The design is MI355X-first: one process
per GPU, hand-written gfx950 HIP kernels on an HBM-resident dataset, RCCL
collectives over xGMI instead of a gRPC parameter server.
"""

__version__ = "0.1.0"

from . import flags  # noqa: F401
from .models.spec import ModelSpec, ArenaLayout, TF_VARIABLE_NAMES  # noqa: F401
