"""Drop-in module tree: same import paths as the reference's ``mivos`` package for the propagation
path, so ``eval_annotation_method.py`` / ``generate_fq_dataset.py`` run unchanged on the HIP engine."""
