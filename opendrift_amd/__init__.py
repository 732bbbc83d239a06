"""opendrift_amd -- MI355X-native particle-advection hot path behind OpenDrift's model/reader API."""
import os as _os

__version__ = '0.2.0'

# Read by the HIP runtime when it initialises (see opendrift_amd/_abi.py: load): more hardware queues than the default 4, so
# that the compute and the upload stream of a context keep queues of their own next to PyTorch's and RCCL's streams.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
# Multi-process GPU work on this platform (RCCL, device memory shared between the ranks of a node) needs the dmabuf IPC mode;
# the launch environment normally exports it already.
_os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
