"""Training-state checkpoints: model + optimizer + sampling stream, for resuming a run.

The reference saves `model.state_dict()` only (causal_cascade/main.py:64-69, vessel_analysis/01_train/train.py:165-182) and has no
resume path; `model.state_dict()` here keeps exactly the reference's keys so those files interchange both ways.  What a
RESUME additionally needs lives outside it: Adam's moments and step count (on the device under HIP-graph replay: FusedAdam.state_dict
writes it back), and the call count of the model's Philox stream (`ops.EpsSource`), without which a resumed run would replay the noise
of its first steps.  Under data parallelism each rank keeps its own stream (different subsequence): save per rank, or save rank 0 and
accept that the other ranks restart their call counts.
"""
import torch


def training_state(model, optimizers=()):
    """dict(model=state_dict (reference keys), optimizers=[...], eps={'calls': n} or None)."""
    if isinstance(optimizers, torch.optim.Optimizer):
        optimizers = (optimizers,)
    eps = getattr(model, "_eps", None)
    return {"model": model.state_dict(), "optimizers": [o.state_dict() for o in optimizers], "eps": eps.state() if eps is not None else None}


def load_training_state(state, model, optimizers=()):
    if isinstance(optimizers, torch.optim.Optimizer):
        optimizers = (optimizers,)
    model.load_state_dict(state["model"])
    if len(state["optimizers"]) != len(optimizers):
        raise ValueError(f"checkpoint holds {len(state['optimizers'])} optimizer states, {len(optimizers)} optimizers given")
    for o, sd in zip(optimizers, state["optimizers"]):
        o.load_state_dict(sd)
    eps = getattr(model, "_eps", None)
    if eps is not None and state.get("eps") is not None:
        eps.load_state(state["eps"])


def save_training_state(path, model, optimizers=()):
    torch.save(training_state(model, optimizers), path)


def resume_training_state(path, model, optimizers=(), map_location=None):
    load_training_state(torch.load(path, map_location=map_location), model, optimizers)
