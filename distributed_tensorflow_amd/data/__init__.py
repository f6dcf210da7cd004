from .sampler import feistel_permute, batch_indices, half_bits  # noqa: F401
from .synthetic import synthetic_mnist  # noqa: F401
from .mnist import DataSet, Datasets, read_data_sets  # noqa: F401
