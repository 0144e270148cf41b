"""Tensor-parallel collectives: the reference's ``torch.distributed`` calls, with the small latency-bound
all-reduces of a decode step optionally served by direct peer reads over xGMI."""
from .xgmi_allreduce import XgmiAllReduce, agree, disable_xgmi_allreduce, enable_xgmi_allreduce, tp_all_gather, tp_all_reduce

__all__ = ["XgmiAllReduce", "enable_xgmi_allreduce", "disable_xgmi_allreduce", "tp_all_reduce", "tp_all_gather", "agree"]
