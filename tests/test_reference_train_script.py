"""The drop-in sentence of BASELINE.json for the TRAINING script: "keeping the mamba_ssm.Mamba / causal_conv1d and
model_segmamba.segmamba.SegMamba module API so it drops into 3_train.py ... unchanged".

The reference's `3_train.py` is executed as it is (everything above its `if __name__ == "__main__":`) with this repository ahead of the
reference tree on sys.path, so that `from model_segmamba.segmamba import SegMamba` inside `BraTSTrainer.__init__` finds the drop-in
package; then ONE iteration of the reference's own loop (`light_training/trainer.py:422-480`: zero the gradients, the script's
`training_step`, backward, `clip_grad_norm_(12)`, `SGD.step()`, logging) runs on a synthetic BraTS-shaped batch.  No GPU here: the
kernels are the library's sources on the CPU emulation (tests/emu), the way every other host-side test runs them.

Needs /root/reference (this container only; skipped on the GPU box) and stand-ins for four third-party imports of the reference that
are not installed in this image and that a single training step never calls: SimpleITK (dataset reader), batchgenerators (file
helpers of the data loader; its augmentation workers feed `train()`, not the step), medpy (validation metric), tensorboard (the
writer gets a recording stub).
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "3_train.py")), reason="reference tree not present")


class _Anything(types.ModuleType):
    """a module whose every attribute exists (a class that accepts anything) - for imports a training step never touches"""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("SimpleITK", "medpy", "tensorboard", "batchgenerators")

    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Anything(spec.name)

    def exec_module(self, module):
        pass


class _Writer:
    def __init__(self, *a, **k):
        self.scalars = []

    def add_scalar(self, k, scalar_value=None, global_step=None):
        self.scalars.append((k, float(scalar_value), global_step))


def test_3_train_py_runs_one_step_of_its_own_loop_on_the_dropin_packages(monkeypatch, tmp_path):
    from tests.emu_util import emu_available, emu_lib
    if not emu_available():
        pytest.skip("no host compiler for the CPU emulation of the kernels")
    from segmamba_amd import lib as L
    monkeypatch.setattr(L, "_lib", emu_lib())
    finder = _StubFinder()
    sys.meta_path.insert(0, finder)
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = _Writer
    monkeypatch.setitem(sys.modules, "torch.utils.tensorboard", tb)
    monkeypatch.syspath_prepend(REF)
    monkeypatch.syspath_prepend(ROOT)                      # the drop-in packages come first
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)          # Trainer.__init__ exits unless num_gpus <= device_count
    monkeypatch.chdir(tmp_path)
    before = set(sys.modules)
    try:
        src = open(os.path.join(REF, "3_train.py")).read()
        head = src.split('if __name__ == "__main__":')[0]
        ns = {"__name__": "train_script", "__file__": os.path.join(REF, "3_train.py")}
        exec(compile(head, os.path.join(REF, "3_train.py"), "exec"), ns)
        import model_segmamba.segmamba as dropin
        assert dropin.__file__.startswith(ROOT)
        trainer = ns["BraTSTrainer"](env_type=ns["env"], max_epochs=1, batch_size=1, device="cpu", logdir=str(tmp_path / "logs"),
                                     val_every=ns["val_every"], num_gpus=ns["num_gpus"], master_port=17759,
                                     training_script=ns["__file__"])
        assert type(trainer.model) is dropin.SegMamba and len(trainer.model.state_dict()) == 291
        assert isinstance(trainer.optimizer, torch.optim.SGD) and trainer.optimizer.defaults["nesterov"]
        g = torch.Generator().manual_seed(0)
        batch = {"data": torch.rand(1, 4, 32, 32, 32, generator=g), "seg": torch.randint(0, 4, (1, 1, 32, 32, 32), generator=g).float()}
        w0 = [p.detach().clone() for p in trainer.model.parameters()]
        with torch.no_grad():
            image, label = trainer.get_input({k: v.clone() for k, v in batch.items()})
            expect = float(torch.nn.functional.cross_entropy(trainer.model(image), label))
        # the state `Trainer.train` sets up in front of its epochs (trainer.py:338-343), then its loop body for one step
        trainer.global_step, trainer.epoch, trainer.writer = 0, 0, _Writer()
        trainer.num_step_per_epoch, trainer.train_loader = 1, iter([batch])
        trainer.train_epoch(0)
        logged = dict((k, v) for k, v, _ in trainer.writer.scalars)
        assert abs(logged["training_loss"] - expect) <= 1e-5 * abs(expect) and logged["lr"] == pytest.approx(1e-2)
        assert trainer.global_step == 1
        moved = [float((p.detach() - w).abs().max()) for p, w in zip(trainer.model.parameters(), w0)]
        assert all(m > 0 for m in moved), "every parameter takes part in the step"
        assert all(torch.isfinite(p).all() for p in trainer.model.parameters())
    finally:
        sys.meta_path.remove(finder)
        for name in set(sys.modules) - before:              # the reference's packages must not leak into other tests
            if name.split(".")[0] in ("light_training", "monai", "SimpleITK", "medpy", "tensorboard", "batchgenerators"):
                sys.modules.pop(name, None)
