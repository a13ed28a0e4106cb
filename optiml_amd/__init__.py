"""MI355X-native box-constrained dual-QP solvers (drop-in for the optiml BCQP path)."""
__version__ = '0.1.0'
