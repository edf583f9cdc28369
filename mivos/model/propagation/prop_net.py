"""``from mivos.model.propagation.prop_net import PropagationNetwork`` -> parameter container whose
``state_dict`` layout matches the reference (stcn.pth loads strictly); compute lives in the HIP engine."""
from eva_vos_amd.params import PropagationNetwork  # noqa: F401
