"""``from mivos.model.fusion_net import FusionNet`` -> parameter container (fusion.pth loads strictly)."""
from eva_vos_amd.params import FusionNet  # noqa: F401
