"""Host view of the device-resident Adam state, in torch.optim.Adam's own state_dict layout.

The reference builds ``optim.Adam(policy.parameters(), lr, eps=1e-5)`` (agents/ppo.py:58), anneals
``param_groups[i]['lr']`` (common/misc_util.py:92-96) and checkpoints ``optimizer.state_dict()``
(agents/ppo.py:271-276, resumed at train.py:260-263).  The moments live in the engine's flat
buffers; a never-stepped torch Adam over the host parameter views is kept purely as the container
that produces / consumes that exact on-disk layout."""
import torch
import torch.optim as optim

from . import layout


class DeviceAdam:
    def __init__(self, policy, engine, lr, eps=1e-5):
        self.policy, self.engine = policy, engine
        self._params = [p for n, p in policy.named_parameters() if not n.startswith("gru.")]
        self._names = [n for n, _ in policy.named_parameters() if not n.startswith("gru.")]
        # the container spans ALL policy parameters like the reference's optim.Adam(policy.parameters()) (a frozen GRU
        # never receives gradients there, so it never gets optimiser state either)
        self._adam = optim.Adam(list(policy.parameters()), lr=lr, eps=eps)
        self.step_count = 0

    @property
    def param_groups(self):
        return self._adam.param_groups

    @property
    def lr(self):
        return float(self._adam.param_groups[0]['lr'])

    def zero_grad(self, set_to_none=True):
        pass                                   # gradients are zeroed inside the fused device step

    def step(self, max_grad_norm, want_norm=False):
        self.step_count += 1
        out = self.engine.optimizer_step(self.lr, max_grad_norm, self.step_count, want_norm)
        self.policy.mark_device_updated()
        return out

    def state_dict(self):
        if self.step_count > 0:
            shapes = self.policy.param_shapes()
            m, v = self.engine.get_adam_state()
            m, v = layout.unflatten(shapes, m), layout.unflatten(shapes, v)
            for n, p in zip(self._names, self._params):
                self._adam.state[p] = {'step': torch.tensor(float(self.step_count)),
                                       'exp_avg': torch.from_numpy(m[n]), 'exp_avg_sq': torch.from_numpy(v[n])}
        return self._adam.state_dict()

    def load_state_dict(self, sd):
        self._adam.load_state_dict(sd)
        shapes = self.policy.param_shapes()
        st = self._adam.state
        if len(st) == 0:
            return
        m = {n: st[p]['exp_avg'].numpy() for n, p in zip(self._names, self._params)}
        v = {n: st[p]['exp_avg_sq'].numpy() for n, p in zip(self._names, self._params)}
        self.engine.set_adam_state(layout.flatten(shapes, m), layout.flatten(shapes, v))
        self.step_count = int(float(st[self._params[0]]['step']))
