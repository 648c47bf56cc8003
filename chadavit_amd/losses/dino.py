"""DINO loss on one fused HIP kernel (reference: src/losses/dino.py:27-118).

One kernel pass per image computes the centred/sharpened teacher softmax, the student log-softmax, the
two cross-view cross-entropies AND dL/dstudent; the centre update (the only explicit collective of the
reference, losses/dino.py:112-114) all-reduces the P-float column sum over RCCL when torch.distributed
is initialised."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from .. import ops


class _DinoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher, center, student_temp, teacher_temp):
        want = student.requires_grad
        loss_rows, dstudent, colsum = ops.dino_loss(student.float().contiguous(), teacher.float().contiguous(),
                                                    center.view(-1).float().contiguous(), student_temp, teacher_temp, want_grad=want)
        ctx.dstudent = dstudent
        ctx.mark_non_differentiable(colsum)
        return loss_rows.mean(), colsum

    @staticmethod
    def backward(ctx, gloss, _gcolsum):
        d, ctx.dstudent = ctx.dstudent, None   # (released with the backward, as autograd releases saved tensors)
        if d is None:
            raise RuntimeError("second backward through the same DINO loss (retain_graph is not supported)")
        return d.float() * gloss, None, None, None, None


class DINOLoss(nn.Module):
    def __init__(self, num_prototypes: int, warmup_teacher_temp: float, teacher_temp: float,
                 warmup_teacher_temp_epochs: float, num_epochs: int, student_temp: float = 0.1, num_large_crops: int = 2,
                 center_momentum: float = 0.9):
        super().__init__()
        self.epoch = 0
        self.student_temp = student_temp
        self.center_momentum = center_momentum
        # `num_large_crops` = the number of views the STUDENT's logits are chunked into (losses/dino.py:82, the reference's name).  The
        # reference's DINO always leaves it at 2 (dino.py:171-178: only the global crops reach the loss); a larger value is the
        # standard-DINO multi-crop loss over 2 teacher x V student views, DINO's `method_kwargs.standard_multicrop_loss` option.
        self.num_large_crops = num_large_crops
        if num_large_crops < 2:
            raise RuntimeError("DINOLoss: at least the 2 global views (the teacher's, losses/dino.py:87) must reach the loss")
        self.register_buffer("center", torch.zeros(1, num_prototypes))
        self.temp_dev = None
        self._pending = None       # (event | work, column sum, 1 / (world * rows)) of a centre update in flight
        self._comm_stream = None
        self.teacher_temp_schedule = np.concatenate((
            np.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs),
            np.ones(num_epochs - warmup_teacher_temp_epochs) * teacher_temp))

    def forward(self, student_output: torch.Tensor, teacher_output: torch.Tensor) -> torch.Tensor:
        # temp_dev: a device float32[1] holding the teacher temperature of this epoch (chadavit_amd.graphed: the captured step reads
        # it from memory instead of freezing the value into the graph)
        temp = self.temp_dev if self.temp_dev is not None else float(self.teacher_temp_schedule[self.epoch])
        # the student's rows are `num_large_crops` views of the teacher's images -- or just the 2 global views (validation_step feeds
        # those alone, dino.py:327-365, whatever the training loss is built over)
        n_img = teacher_output.shape[0] // 2
        views = student_output.shape[0] // max(n_img, 1)
        if n_img * 2 != teacher_output.shape[0] or views * n_img != student_output.shape[0] or views not in (2, self.num_large_crops):
            raise RuntimeError(f"DINOLoss: {student_output.shape[0]} student rows are not {self.num_large_crops} views (or the 2 global "
                               f"views) of the {n_img} images the teacher saw")
        self.sync_center()  # the previous step's centre update (its all-reduce ran beside that step's backward)
        loss, colsum = _DinoLossFn.apply(student_output, teacher_output, self.center, float(self.student_temp), temp)
        self.update_center(teacher_output, colsum)
        return loss

    @torch.no_grad()
    def update_center(self, teacher_output: torch.Tensor, colsum: torch.Tensor = None):
        """c <- m c + (1-m) * sum_rows(t) [all-reduce SUM] / world / rows   (losses/dino.py:103-118).

        The new centre is first read by the NEXT step's loss, so with several ranks the P-float all-reduce is
        only STARTED here -- on the communication stream (RCCL) or as an asynchronous gloo work -- and `sync_center` finishes
        the update (wait + the EMA kernel) right before the centre is read again: the collective is no rank rendezvous between
        the loss and the backward, and the centre each rank ends up with is the one the blocking form produces."""
        if colsum is None:
            colsum = ops.sum_rows_f32(teacher_output.float().contiguous())
        self.sync_center()
        inv = 1.0 / len(teacher_output)
        from ..parallel import force_collectives
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives()):
            inv /= dist.get_world_size()
            if dist.get_backend() == "nccl":
                main = torch.cuda.current_stream(colsum.device)
                if self._comm_stream is None:
                    self._comm_stream = torch.cuda.Stream(device=colsum.device)
                self._comm_stream.wait_stream(main)  # the column sum is complete on the compute stream
                with torch.cuda.stream(self._comm_stream):
                    dist.all_reduce(colsum)
                    done = torch.cuda.Event()
                    done.record(self._comm_stream)
                colsum.record_stream(self._comm_stream)
                self._pending = (done, colsum, inv)
            else:
                self._pending = (dist.all_reduce(colsum, async_op=True), colsum, inv)
            return
        ops.center_ema(self.center.view(-1), colsum, inv, float(self.center_momentum))

    @torch.no_grad()
    def sync_center(self):
        """Finish a centre update whose all-reduce is still in flight (no-op otherwise).  Called before every read of `center`
        by this module (forward, state_dict); call it before reading the buffer directly."""
        if self._pending is None:
            return
        done, colsum, inv = self._pending
        self._pending = None
        if isinstance(done, torch.cuda.Event):
            torch.cuda.current_stream(colsum.device).wait_event(done)
        else:
            done.wait()
        ops.center_ema(self.center.view(-1), colsum, inv, float(self.center_momentum))

    def _save_to_state_dict(self, destination, prefix, keep_vars):  # (also reached through a parent module's state_dict())
        self.sync_center()
        super()._save_to_state_dict(destination, prefix, keep_vars)
