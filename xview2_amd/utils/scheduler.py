"""Noam learning-rate schedule (linear warm-up, exponential decay), stepped once per optimizer step.
Host-side scalar arithmetic with the constructor contract of the reference's utils/scheduler.py:6-59
(warmup_epochs, total_epochs, steps_per_epoch, init_lr, max_lr, final_lr); works with any optimizer object
exposing ``param_groups`` (torch.optim.* or xview2_amd.optim.FlatAdamW)."""


class NoamLR:
    def __init__(self, optimizer, warmup_epochs, total_epochs, steps_per_epoch, init_lr, max_lr, final_lr):
        self.optimizer = optimizer
        self.n = len(optimizer.param_groups)
        self.init_lr, self.max_lr, self.final_lr = init_lr, max_lr, final_lr
        self.warmup_steps = int(warmup_epochs * steps_per_epoch)
        self.total_steps = total_epochs * steps_per_epoch
        # --warmup 0: the reference's numpy division yields inf with a warning and the branch is never taken
        self.linear_increment = (max_lr - init_lr) / self.warmup_steps if self.warmup_steps else 0.0
        decay_steps = self.total_steps - self.warmup_steps
        # no decay phase (--epochs == --warmup): the reference's numpy power gives 0.0 with a warning, never used
        self.exponential_gamma = (final_lr / max_lr) ** (1 / decay_steps) if decay_steps else 0.0
        self.current_step = 0
        self.lr = [init_lr] * self.n
        # torch's _LRScheduler.__init__ (the reference's base class, utils/scheduler.py:43) performs one
        # step() at construction: the schedule starts at current_step == 1
        self.step()

    def get_lr(self):
        return list(self.lr)

    def step(self, current_step=None):
        self.current_step = current_step if current_step is not None else self.current_step + 1
        s = self.current_step
        if s <= self.warmup_steps:
            lr = self.init_lr + s * self.linear_increment
        elif s <= self.total_steps:
            lr = self.max_lr * (self.exponential_gamma ** (s - self.warmup_steps))
        else:
            lr = self.final_lr
        for i, g in enumerate(self.optimizer.param_groups):
            self.lr[i] = lr
            g["lr"] = lr
