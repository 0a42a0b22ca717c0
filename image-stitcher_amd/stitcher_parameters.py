"""Stitching parameters: the configuration surface the front-ends hand to ``Stitcher``.

Mirrors the reference's dataclass (stitcher_parameters.py:8-107): same field names,
defaults, validation errors, ``stitched_folder`` naming and JSON round-trip, so a
params JSON written for the reference loads here unchanged.
"""
from __future__ import annotations

import dataclasses
import json
import os
from datetime import datetime
from typing import Any, Dict

_FORMATS = ('.ome.zarr', '.ome.tiff')
_PATTERNS = ('Unidirectional', 'S-Pattern')


@dataclasses.dataclass
class StitchingParameters:
    input_folder: str
    output_format: str = '.ome.zarr'
    apply_flatfield: bool = False
    use_registration: bool = False
    registration_channel: str = ''      # empty -> first channel in sorted order
    registration_z_level: int = 0
    dynamic_registration: bool = False  # parsed and stored; the reference never reads it
    scan_pattern: str = 'Unidirectional'
    merge_timepoints: bool = False
    merge_hcs_regions: bool = False

    def __post_init__(self) -> None:
        self.input_folder = os.path.abspath(self.input_folder)
        if self.registration_channel is None:   # argparse default when the flag is absent
            self.registration_channel = ''

    def validate(self) -> None:
        """ValueError on a missing folder, unknown format/pattern or negative z
        (reference stitcher_parameters.py:36-59)."""
        if not os.path.exists(self.input_folder):
            raise ValueError(f"Input folder does not exist: {self.input_folder}")
        if self.output_format not in _FORMATS:
            raise ValueError("Output format must be either .ome.zarr or .ome.tiff")
        if self.scan_pattern not in _PATTERNS:
            raise ValueError("Scan pattern must be either 'Unidirectional' or 'S-Pattern'")
        if self.use_registration and self.registration_z_level < 0:
            raise ValueError("Registration Z-level must be non-negative")

    @property
    def stitched_folder(self) -> str:
        """``{input}_stitched_{timestamp}`` (reference stitcher_parameters.py:61-64)."""
        stamp = datetime.now().strftime('%Y-%m-%d_%H-%M-%S.%f')
        return self.input_folder + "_stitched_" + stamp

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> 'StitchingParameters':
        names = {f.name for f in dataclasses.fields(cls)}
        return cls(**{k: v for k, v in data.items() if k in names})

    @classmethod
    def from_json(cls, json_path: str) -> 'StitchingParameters':
        with open(json_path) as fh:
            return cls.from_dict(json.load(fh))

    def to_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)

    def to_json(self, json_path: str) -> None:
        with open(json_path, 'w') as fh:
            json.dump(self.to_dict(), fh, indent=2)
