from xitorch_amd.grad.jachess import jac, hess

__all__ = ["jac", "hess"]
