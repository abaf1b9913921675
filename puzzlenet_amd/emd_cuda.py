"""Drop-in for the reference's compiled extension module ``emd_cuda``
(PyTorchEMD/cuda/emd.cpp:23-27): the three pybind names, same argument order.
"""
from . import ops

approxmatch_forward = ops.emd_approxmatch        # (xyz1[B,n,3], xyz2[B,m,3]) -> match[B,m,n]
matchcost_forward = ops.emd_matchcost            # (xyz1, xyz2, match) -> cost[B]
matchcost_backward = ops.emd_matchcost_grad      # (grad_cost, xyz1, xyz2, match) -> [grad1, grad2]
