from xitorch_amd.optimize.rootfinder import rootfinder, equilibrium, minimize

__all__ = ["rootfinder", "equilibrium", "minimize"]
