"""Drop-in for code/sync_batchnorm: one process per GPU, statistics all-reduced by RCCL (gan_ops.SynchronizedBatchNorm2d);
DataParallelWithCallback is the identity because there is no single-process replication to patch."""
import importlib

from _m355 import pkg as _pkg  # noqa: F401

SynchronizedBatchNorm2d = importlib.import_module("2dimageto3dmodel_amd.gan_ops").SynchronizedBatchNorm2d


def DataParallelWithCallback(module, device_ids=None):
    return module
