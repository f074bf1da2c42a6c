"""MI355X-native GE2E speaker-embedding hot path (hand-written gfx950 kernels behind a C ABI).

Host-side mirror of the reference's module layout: `Modules` (GE2E, GE2E_Loss), `distributed`,
`Arg_Parser`, `Train` (Trainer), `Inference` (Inferencer)."""
from .Modules import GE2E, GE2E_Loss, GE2E_Loss_Global  # noqa: F401

__version__ = "0.1.0"
