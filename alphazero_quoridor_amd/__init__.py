"""alphazero_quoridor_amd -- MI355X-native self-play engine for 9x9 Quoridor.

Drop-in for the self-play data-generation path of cryer/AlphaZero_Quoridor: the rules
engine and the MCTS select / expand / backup loop are HIP kernels (``csrc/``) behind a C
ABI (``include/qz_abi.h``).  The policy-value network's WEIGHTS, training and the library
route of its forward pass stay in PyTorch-ROCm (``policy_value_net.py``); on the engine
route -- what self-play and ``bench.py`` run -- a leaf evaluation is ``qz_nn_evaluate``:
the unchanged architecture as two HIP launches (``csrc/qz_conv.hip``, ``qz_nn.hip``),
within 1e-5 of the reference's ``policy_value_fn``.  The modules
``quoridor``, ``mcts``, ``policy_value_net`` and ``train`` mirror the reference's module
surface (same class / method names, argument meaning and return conventions).
"""
__version__ = "0.1.0"

from . import _cabi  # noqa: F401
