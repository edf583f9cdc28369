"""eva_vos_amd - MI355X-native STCN mask-propagation engine (drop-in for mivos.inference_core)."""
__version__ = "0.1.0"
