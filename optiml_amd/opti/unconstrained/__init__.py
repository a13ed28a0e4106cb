"""Unconstrained first-order optimizers — only what the augmented-Lagrangian dual path needs:
the full-batch "stochastic" update rules, device-resident.  Line-search methods and the proximal bundle of
optiml/opti/unconstrained are out of scope."""
__all__ = ['stochastic']

from . import stochastic
