"""alphazero_quoridor_amd -- MI355X-native self-play engine for 9x9 Quoridor.

Drop-in for the self-play data-generation path of cryer/AlphaZero_Quoridor: the rules
engine and the MCTS select / expand / backup loop are HIP kernels (``csrc/``) behind a C
ABI (``include/qz_abi.h``); the policy-value network stays in PyTorch-ROCm.  The modules
``quoridor``, ``mcts``, ``policy_value_net`` and ``train`` mirror the reference's module
surface (same class / method names, argument meaning and return conventions).
"""
__version__ = "0.1.0"

from . import _cabi  # noqa: F401
