"""``from mivos.inference_core import InferenceCore`` -> HIP engine (see eva_vos_amd/inference_core.py)."""
from eva_vos_amd.inference_core import InferenceCore  # noqa: F401
