"""Import-path compatibility with the reference tree: ``model.unet``, ``model.layers``, ``model.loss`` and
``model.plt`` resolve to the HIP implementations (xview2_amd.networks / decoder / criterion / lightning)."""
