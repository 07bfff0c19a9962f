"""Checkpoint wire format of the reference (util.py:124-163): one torch.save'd dict

    {epoch, iter, graph: graph.state_dict(), optim, optim_pose, sched, sched_pose}

with the reference's state-dict keys (nerf.mlp_feat.N.weight ..., nerf.progress, [nerf_fine.*],
warp_mlp.lin{b}_{a,b}_0.{weight_g,weight_v,bias}, warp_mlp.lin{b}_{a,b}_1.*, warp_mlp.lin{b}_c.*,
warp_latent.weight, global_rigid.weight) and optimizer / scheduler states in torch.optim's own format, laid
out as the reference's setup_optimizer builds them (nerf_inn_llff.py:34-47: `optim` = Adam, group 0 nerf,
group 1 nerf_fine; barf_inn_llff.py:84-104: `optim_pose` = Adam, group 0 warp_mlp, group 1 warp_latent; both
under ExponentialLR).  A reference-trained model.ckpt therefore restores into the INNTrainer (weights, Adam
moments, step, learning-rate schedule) and vice versa.

The engine keeps Adam moments as flat buffers (one fused niw_adam_step per group); here they are scattered
to / gathered from per-parameter `exp_avg` / `exp_avg_sq` entries through real torch.optim objects that are
used only as (de)serialisers -- they never step.
"""
import os
import shutil

import torch


def group_by_child(state_dict):
    """`graph` section of a checkpoint -> {child module name: its own state dict}: one pass over the keys, split at the first dot
    (what the reference does per child with util.get_child_state_dict, util.py:120-121)"""
    groups = {}
    for key, value in state_dict.items():
        child, _, rest = key.partition(".")
        if rest:
            groups.setdefault(child, {})[rest] = value
    return groups


def get_child_state_dict(state_dict, key):
    return group_by_child(state_dict).get(key, {})


def checkpoint_path(opt, resume=True):
    """the file `--resume` names: the latest checkpoint, or the numbered copy of epoch / iteration `resume` (util.py:126-128)"""
    if resume is True:
        return os.path.join(opt.output_path, "model.ckpt")
    return os.path.join(opt.output_path, "model", f"{resume}.ckpt")


def _optimizers(trainer):
    """torch.optim.Adam / ExponentialLR shells with the reference's group layout -> dict(optim, optim_pose,
    sched, sched_pose) plus, per optimizer, the trainer group index behind each param group."""
    opt = trainer.opt
    n_nets = len(trainer.nets)
    o = opt.optim
    optim = torch.optim.Adam([dict(params=list(net.parameters()), lr=o.lr) for net in trainer.nets])
    if getattr(trainer, "family", None) == "vanilla":
        # the vanilla model (reference model/nerf.py:34-47): ONE optimizer, one param group per network, one scheduler
        sched = torch.optim.lr_scheduler.ExponentialLR(optim, gamma=trainer.gammas[0])
        return dict(optim=optim, sched=sched), dict(optim=list(range(n_nets)))
    # optim_pose: group 0 = warp network, group 1 = latent table; a group the options switch off
    # (inn.optimize.enabled / warp_latent.optimize.enabled, barf_inn_llff.py:88-93) is absent, as in the reference
    pose_modules = [(n_nets, trainer.warp_mlp), (n_nets + 1, trainer.warp_latent)]
    pose_modules = [(gi, m) for gi, m in pose_modules if trainer.trainable[gi]]
    optim_pose = torch.optim.Adam([dict(params=list(m.parameters()), lr=o.lr_pose) for _, m in pose_modules])
    sched = torch.optim.lr_scheduler.ExponentialLR(optim, gamma=trainer.gammas[0])
    sched_pose = torch.optim.lr_scheduler.ExponentialLR(optim_pose, gamma=trainer.gammas[n_nets])
    return dict(optim=optim, optim_pose=optim_pose, sched=sched, sched_pose=sched_pose), \
        dict(optim=list(range(n_nets)), optim_pose=[gi for gi, _ in pose_modules])


def _param_slices(trainer, gi):
    """{id(param): (offset, numel)} of trainer group gi inside its flat buffers"""
    out, off = {}, 0
    for p in trainer.bucket.groups[gi]:
        out[id(p)] = (off, p.numel())
        off += p.numel()
    return out


def optimizer_state_dicts(trainer):
    """-> {optim, optim_pose, sched, sched_pose} state dicts in torch format for the trainer's current step"""
    objs, groups = _optimizers(trainer)
    it = trainer.it
    for name in [n for n in ("optim", "optim_pose") if n in objs]:
        optim = objs[name]
        for pg, gi in zip(optim.param_groups, groups[name]):
            where = _param_slices(trainer, gi)
            lr0 = trainer.lrs[gi][0]
            pg["lr"] = lr0 * trainer.gammas[gi] ** it
            pg["initial_lr"] = lr0
            if it == 0:
                continue
            for p in pg["params"]:
                if id(p) not in where:                       # e.g. nerf.progress: never receives a gradient
                    continue
                off, n = where[id(p)]
                optim.state[p] = dict(step=torch.tensor(float(it)),
                                      exp_avg=trainer.m[gi][off:off + n].view(p.shape).clone(),
                                      exp_avg_sq=trainer.v[gi][off:off + n].view(p.shape).clone())
    out = {k: objs[k].state_dict() for k in ("optim", "optim_pose") if k in objs}
    for name in [n for n in ("sched", "sched_pose") if n in objs]:
        s = objs[name]
        s.last_epoch, s._step_count = it, it + 1
        s._last_lr = [pg["lr"] for pg in s.optimizer.param_groups]
        out[name] = s.state_dict()
    return out


def load_optimizer_state_dicts(trainer, checkpoint):
    """Scatter `optim` / `optim_pose` Adam moments of a reference-format checkpoint into the trainer's flat
    buffers (parameters without state keep zero moments)."""
    objs, groups = _optimizers(trainer)
    for name in ("optim", "optim_pose"):
        if name not in checkpoint or name not in objs:
            continue
        optim = objs[name]
        optim.load_state_dict(checkpoint[name])
        for pg, gi in zip(optim.param_groups, groups[name]):
            where = _param_slices(trainer, gi)
            trainer.m[gi].zero_()
            trainer.v[gi].zero_()
            for p in pg["params"]:
                st = optim.state.get(p)
                if not st or id(p) not in where:
                    continue
                off, n = where[id(p)]
                trainer.m[gi][off:off + n].copy_(st["exp_avg"].reshape(-1))
                trainer.v[gi][off:off + n].copy_(st["exp_avg_sq"].reshape(-1))


def save_checkpoint(opt, trainer, ep, it, latest=False, children=None):
    """reference util.py:147-163; `trainer` plays the reference's `model` (has .graph and optimizer state).  Issues NO collective:
    under ray sharding the caller runs `trainer.sync_state()` on EVERY rank first (Model.save_checkpoint does) and only then lets
    rank 0 write -- a collective behind a rank gate would pair with the other ranks' next gradient all-reduce."""
    os.makedirs("{0}/model".format(opt.output_path), exist_ok=True)
    if hasattr(trainer, "sync_state"):
        trainer.sync_state(collective=False)
    sd = trainer.graph.state_dict()
    if children is not None:
        sd = {k: v for k, v in sd.items() if k.startswith(children)}
    checkpoint = dict(epoch=ep, iter=it, graph=sd)
    checkpoint.update(optimizer_state_dicts(trainer))
    torch.save(checkpoint, "{0}/model.ckpt".format(opt.output_path))
    if not latest:
        shutil.copy("{0}/model.ckpt".format(opt.output_path), "{0}/model/{1}.ckpt".format(opt.output_path, ep or it))


def restore_checkpoint(opt, trainer, load_name=None, resume=False):
    """reference util.py:124-145 -> (epoch, iter) when resuming, (None, None) when only loading weights"""
    if (load_name is None) == (resume is False):
        raise ValueError("restore_checkpoint: give either load_name (weights only) or resume (weights, optimizers, counters)")
    if resume:
        load_name = checkpoint_path(opt, resume)
    checkpoint = torch.load(load_name, map_location=opt.device, weights_only=False)
    per_child = group_by_child(checkpoint["graph"])
    for name, child in trainer.graph.named_children():
        if name in per_child:
            child.load_state_dict(per_child[name])
    for net in trainer.nets:                                   # host copy of the c2f progress the kernels use
        if hasattr(net, "progress"):
            net.set_progress(float(net.progress.data))
    if not resume:
        return None, None
    load_optimizer_state_dicts(trainer, checkpoint)
    ep, it = checkpoint["epoch"], checkpoint["iter"]
    if resume is not True:
        assert resume == (ep or it)
    trainer.it = it or 0
    return ep, it
