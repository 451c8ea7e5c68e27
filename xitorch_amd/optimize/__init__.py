from xitorch_amd.optimize.rootfinder import rootfinder

__all__ = ["rootfinder"]
