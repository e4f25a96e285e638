"""Mirror of pytorch/FasterRCNN/utils.py:12-16 (the no_grad decorator used by predict)."""
import torch as t


def no_grad(func):
    def wrapper_nograd(*args, **kwargs):
        with t.no_grad():
            return func(*args, **kwargs)
    return wrapper_nograd
