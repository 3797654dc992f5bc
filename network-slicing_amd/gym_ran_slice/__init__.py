from ranslice.gymshim import register

from .ran_slice import RanSlice  # noqa: F401

register(id='RanSlice-v1', entry_point='gym_ran_slice:RanSlice')
