"""Which part of the library backward (MIOpen convolutions / batch norm) is responsible for the distance
between the GPU's fp32 gradients and the float64 gradients of the train fixture?"""
import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from _stubs import det_fill_state_dict
from alphazero_quoridor_amd import rules
from alphazero_quoridor_amd.boards import DeviceBoards
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

dev = torch.device("cuda:0")
d = np.load(os.path.join(ROOT, "tests", "golden", "train_fixture.npz"))
keys = [k[4:] for k in d.files if k.startswith("g64_")]

def grads(flags, label):
    pvn = PolicyValueNet(use_gpu=True, device=dev)
    pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), int(d["fill_seed"])))
    states = rules.encode(DeviceBoards.from_packed(d["board"], dev))
    pi, z = torch.from_numpy(d["pi"]).to(dev), torch.from_numpy(d["z"]).to(dev)
    with torch.backends.cudnn.flags(**flags):
        pvn.optimizer.zero_grad()
        logp, v = pvn.policy_value_net(states)
        loss = F.mse_loss(v.view(-1), z) - torch.mean(torch.sum(pi * logp, 1))
        loss.backward()
    named = dict(pvn.policy_value_net.named_parameters())
    out = []
    for k in keys:
        name = [n for n in named if n.replace(".", "_") == k][0]
        g = named[name].grad.cpu().numpy().astype(np.float64)
        out.append(np.abs(g - d["g64_" + k]).max() / np.abs(d["g64_" + k]).max())
    print("%-44s " % label + " ".join("%.1e" % e for e in out), flush=True)

print(" " * 45 + " ".join("%-7s" % k[:7] for k in keys))
grads(dict(enabled=True, benchmark=False, deterministic=False), "MIOpen, default")
grads(dict(enabled=True, benchmark=False, deterministic=False), "MIOpen, default (again)")
grads(dict(enabled=True, benchmark=True, deterministic=False), "MIOpen, benchmark=True")
grads(dict(enabled=True, benchmark=False, deterministic=True), "MIOpen, deterministic=True")
grads(dict(enabled=False), "backends.cudnn disabled (native kernels)")

# the same module in float64 on the GPU: separates "the module is not the reference's" from "fp32 noise"
pvn = PolicyValueNet(use_gpu=True, device=dev)
pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), int(d["fill_seed"])))
pvn.policy_value_net.double()
states = rules.encode(DeviceBoards.from_packed(d["board"], dev)).double()
pi, z = torch.from_numpy(d["pi"]).to(dev).double(), torch.from_numpy(d["z"]).to(dev).double()
logp, v = pvn.policy_value_net(states)
loss = F.mse_loss(v.view(-1), z) - torch.mean(torch.sum(pi * logp, 1))
loss.backward()
named = dict(pvn.policy_value_net.named_parameters())
out = []
for k in keys:
    name = [n for n in named if n.replace(".", "_") == k][0]
    out.append(np.abs(named[name].grad.cpu().numpy() - d["g64_" + k]).max() / np.abs(d["g64_" + k]).max())
print("%-44s " % "float64 on the GPU" + " ".join("%.1e" % e for e in out), "loss", float(loss))
# fp32 on the CPU with OUR module (the reference's numbers came from ITS module)
cpu = PolicyValueNet(use_gpu=False)
cpu.policy_value_net.load_state_dict(det_fill_state_dict(cpu.policy_value_net.state_dict(), int(d["fill_seed"])))
st = rules.encode(DeviceBoards.from_packed(d["board"], dev)).cpu()
logp, v = cpu.policy_value_net(st)
loss = F.mse_loss(v.view(-1), torch.from_numpy(d["z"])) - torch.mean(torch.sum(torch.from_numpy(d["pi"]) * logp, 1))
loss.backward()
named = dict(cpu.policy_value_net.named_parameters())
out = []
for k in keys:
    name = [n for n in named if n.replace(".", "_") == k][0]
    out.append(np.abs(named[name].grad.numpy().astype(np.float64) - d["g64_" + k]).max() / np.abs(d["g64_" + k]).max())
print("%-44s " % "our module, fp32 on the CPU" + " ".join("%.1e" % e for e in out))
