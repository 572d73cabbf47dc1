"""lightning_amd: MI355X-native batched secp256k1 signature verification behind Core Lightning's
check_signed_hash / check_tx_sig / check_schnorr_sig / gossipd sigcheck_* interface.

The product is liblightning_amd.so (C ABI: include/lightning_amd.h; kernels: csrc/).  This
package is the thin Python front end used by tests and bench.py."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before the HIP runtime starts: the engine's lanes need their own hardware queues
from .engine import Engine, LamdError  # noqa: F401

__all__ = ["Engine", "LamdError"]
