"""Mirror of the reference's ``models`` package (same module and class names), running on the gfx950 kernels."""
