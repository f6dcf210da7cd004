"""distributed_tensorflow_amd: an MI355X-native distributed deep-MNIST trainer.

Capabilities mirror ellie-ba/Distributed_TensorFlow (reference
.idea/MNISTDist.py): the deep-MNIST CNN, the ps/worker flag surface, chief
managed tf.train.Saver-compatible checkpoints, async parameter-server and
synchronous data-parallel training.  The design is MI355X-first: one process
per GPU, hand-written gfx950 HIP kernels on an HBM-resident dataset, RCCL
collectives over xGMI instead of a gRPC parameter server.
"""

__version__ = "0.1.0"

import os as _os

# Peer mappings (hipIpc handles of the exchange buffers, RCCL's own) need the dmabuf IPC mode on this driver stack.
# The HSA runtime reads its environment ONCE, when the first HIP call initialises it, so the variable is set here, at
# package import -- which every entry point (mnistdist.py, bench.py, the test workers) does before it touches the GPU --
# and not next to init_process_group, where the runtime is already up (that line was a no-op: round-2 review).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import flags  # noqa: F401
from .models.spec import ModelSpec, ArenaLayout, TF_VARIABLE_NAMES  # noqa: F401
