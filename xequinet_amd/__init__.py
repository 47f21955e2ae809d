"""xequinet_amd -- MI355X-native XPaiNN energy+force hot path (drop-in for the
corresponding pieces of X1X1010/XequiNet).  HIP kernels live in csrc/ behind the
C ABI of include/xeq.h; this package is the host-side mirror of the reference's
``nn.Module`` / functional interface.  There is no CPU fallback."""
from . import keys  # noqa: F401

__version__ = "0.1.0"
