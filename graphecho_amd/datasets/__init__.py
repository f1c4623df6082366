"""On-disk decoders in front of the input-formatting kernels: host I/O only."""
from .camus import CamusSet
from .cardiac_uda import CardiacUDASet
from .echo import EchoFrames, EchoSet
from .formats import bgr_to_gray, read_avi_mjpeg, read_mhd, read_nifti, write_avi_mjpeg, write_mhd, write_nifti
from .loader import RawBatches
from .raster import fill_poly, polygon

__all__ = ["CamusSet", "CardiacUDASet", "EchoSet", "EchoFrames", "RawBatches", "read_mhd", "read_nifti", "write_mhd", "write_nifti",
           "read_avi_mjpeg", "write_avi_mjpeg", "bgr_to_gray", "polygon", "fill_poly"]
