"""Drop-in mirror of the reference's operator surface ``manner/models/components``:
same class names, constructor arguments, forward signatures and state_dict keys; forward runs the
gfx950 HIP hot path."""
