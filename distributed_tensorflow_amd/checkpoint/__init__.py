from .bundle import CheckpointError, list_bundle, read_bundle, write_bundle  # noqa: F401
from .saver import Saver, from_tf_names, latest_checkpoint, read_checkpoint_state, to_tf_names  # noqa: F401
