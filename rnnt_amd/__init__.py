"""rnnt_amd: RNN-Transducer loss for AMD MI355X (gfx950).

Public surface (mirrors the loss call surface of the reference model):
  rnnt_loss(acts, labels, input_lengths, label_lengths, blank_label)  -> costs[B]
  rnnt_joint_loss(f, g, labels, input_lengths, label_lengths, blank)    -> costs[B]  (additive joint, fused)
  rnnt_lambda_func / rnnt_lambda_func_v2                              -> costs[B]
  concatenate_lambda / fc_joint / fc_joint_loss                       (FC-joint model front end)
  joint_log_softmax / softmax                                         (decoder inner step, batched)
  pad_batch                                                           (the batch producer's array layout)
"""
from .loss import RNNTLoss, rnnt_joint_loss, rnnt_loss, rnnt_loss_grad_  # noqa: F401
from .lambdas import rnnt_lambda_func, rnnt_lambda_func_v2  # noqa: F401
from .fc_joint import concatenate_lambda, fc_joint, fc_joint_first_layer, fc_joint_loss, joint_dense  # noqa: F401
from .decode import joint_log_softmax, softmax  # noqa: F401
from .batching import pad_batch  # noqa: F401
from .sharding import partition_lpt, shard_batch, allreduce_scalars  # noqa: F401

__all__ = [
    "rnnt_loss", "rnnt_joint_loss", "rnnt_loss_grad_", "RNNTLoss", "rnnt_lambda_func", "rnnt_lambda_func_v2",
    "concatenate_lambda", "fc_joint", "fc_joint_first_layer", "fc_joint_loss", "joint_dense",
    "joint_log_softmax", "softmax", "partition_lpt", "shard_batch", "allreduce_scalars", "pad_batch",
]
