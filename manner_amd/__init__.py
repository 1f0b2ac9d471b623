"""MI355X-native hot path of andreeaiana/manner: news encoding + candidate scoring behind the reference's own operator surface.

``manner_amd.install()`` binds the HIP classes into the reference's ``manner.models.components`` modules (manner_amd/binding.py);
everything else is imported from the submodules (``hip``, ``hotpath``, ``distributed``, ``train``, ``models.components``)."""


def install(reference_root=None):
    from .binding import install as _install
    return _install(reference_root)


def uninstall():
    from .binding import uninstall as _uninstall
    return _uninstall()
