"""`from zipnn.zipnn import …` (reference zipnn/zipnn.py) → zipnn_amd.zipnn."""
from zipnn_amd.zipnn import *  # noqa: F401,F403
from zipnn_amd.zipnn import ZipNN, SafeOpen, zipnn_hf, zipnn_safetensors, decompress_safetensors_tensor  # noqa: F401
