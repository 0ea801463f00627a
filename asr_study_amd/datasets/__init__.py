from .dataset_parser import DatasetParser  # noqa: F401
from .dummy import Dummy  # noqa: F401
from .dataset_generator import DatasetGenerator, DatasetIterator, H5Iterator  # noqa: F401
