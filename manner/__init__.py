"""Import-path shim: ``manner.models.components.*`` as the reference's LightningModules import them
(reference manner/models/cr_module.py:13-16, a_module.py:15, ensemble_module.py:13), re-exporting the MI355X mirror
classes of ``manner_amd``.  Dropped next to (or over) the reference's own four component files, the reference's
``CRModule`` / ``AModule`` / ``EnsembleModule`` and Hydra configs run on the HIP hot path unchanged."""
