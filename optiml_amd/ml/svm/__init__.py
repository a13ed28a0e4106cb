__all__ = ['SVM', 'SVC', 'SVR']

from ._base import SVM, SVC, SVR
