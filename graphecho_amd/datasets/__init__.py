"""On-disk decoders in front of the input-formatting kernels: host I/O only."""
from .camus import CamusSet
from .cardiac_uda import CardiacUDASet
from .formats import read_mhd, read_nifti, write_mhd, write_nifti
from .loader import RawBatches

__all__ = ["CamusSet", "CardiacUDASet", "RawBatches", "read_mhd", "read_nifti", "write_mhd", "write_nifti"]
