"""Class registry with the reference's table and key names.

Mirrors the plug-in mechanism of funasr/register.py:7-92 (`tables.register(table, key)` decorator, lookup by
`tables.<table>.get(name)`, re-registering a key overrides it, :63-68). When the real `funasr` package is importable
`funasr_amd.install()` (see funasr_amd/install.py) registers the same classes into the reference's own `tables`, which
is the drop-in route; this module is what the stand-alone AutoModel shim uses when `funasr` is not installed.
"""
from __future__ import annotations

import inspect
import logging

TABLE_NAMES = ("model_classes", "frontend_classes", "specaug_classes", "normalize_classes", "encoder_classes",
               "decoder_classes", "joint_network_classes", "predictor_classes", "stride_conv_classes",
               "tokenizer_classes", "dataloader_classes", "batch_sampler_classes", "dataset_classes",
               "index_ds_classes")


class RegisterTables:
    def __init__(self):
        for t in TABLE_NAMES:
            setattr(self, t, {})
            setattr(self, t + "_meta", {})

    def register(self, register_tables_key: str, key: str | None = None):
        def decorator(target_class):
            if not hasattr(self, register_tables_key):
                setattr(self, register_tables_key, {})
                setattr(self, register_tables_key + "_meta", {})
            registry = getattr(self, register_tables_key)
            registry_key = key if key is not None else target_class.__name__
            if registry_key in registry:
                logging.debug("Key %s already exists in %s, re-register", registry_key, register_tables_key)
            registry[registry_key] = target_class
            try:
                where = f"{inspect.getfile(target_class)}:{inspect.getsourcelines(target_class)[1]}"
            except (OSError, TypeError):
                where = "<unknown>"
            getattr(self, register_tables_key + "_meta")[registry_key] = [registry_key, target_class.__name__, where]
            return target_class

        return decorator

    def print(self, key: str | None = None) -> None:
        for t in TABLE_NAMES:
            meta = getattr(self, t + "_meta", {})
            if meta and (key is None or key in t):
                print(f"-----------    ** {t} **    --------------")
                for row in sorted(meta.values()):
                    print("| " + " | ".join(str(x) for x in row) + " |")


tables = RegisterTables()
