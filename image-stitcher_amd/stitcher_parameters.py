"""The configuration object the front-ends hand to ``Stitcher``.

Field names, defaults, validation messages, the ``stitched_folder`` naming and the JSON round trip
follow the reference's dataclass (stitcher_parameters.py:8-107), so a params JSON written for the
reference loads here unchanged.
"""
from __future__ import annotations

import dataclasses
import json
import os
from datetime import datetime
from typing import Any, Dict

OUTPUT_FORMATS = ('.ome.zarr', '.ome.tiff')
SCAN_PATTERNS = ('Unidirectional', 'S-Pattern')


@dataclasses.dataclass
class StitchingParameters:
    """What each field does on the MI355X path:

    ==========================  =========================================================================
    ``input_folder``            Squid acquisition: ``acquisition parameters.json`` plus one sub-folder per
                                timepoint holding ``coordinates.csv`` and ``<region>_<fov>_<z>_<channel>`` tiles
    ``output_format``           ``.ome.zarr`` (zarr v2 + NGFF 0.4, written plane by plane, also by several
                                ranks at once) or ``.ome.tiff`` (BigTIFF with OME-XML)
    ``apply_flatfield``         tiles are divided by their channel's gain image inside the fusion kernel
    ``use_registration``        the centre tile and its right / bottom neighbours are phase-correlated on the
                                device and every tile is placed by the two measured shifts; otherwise tiles are
                                placed by their stage coordinates
    ``registration_channel``    channel the shifts are measured on
    ``registration_z_level``    z plane the shifts are measured on
    ``dynamic_registration``    parsed, stored and ignored -- exactly what the reference does with it
                                (stitcher.py:92); all-pairs registration is ``Stitcher(all_pairs_registration=True)``
    ``scan_pattern``            ``S-Pattern`` adds a third pair so that reversed stage rows get their own shift
    ``merge_timepoints`` /      output re-packaging requests; the per-(timepoint, region) stores are always
    ``merge_hcs_regions``       written
    ==========================  =========================================================================
    """
    input_folder: str
    output_format: str = OUTPUT_FORMATS[0]
    apply_flatfield: bool = False
    use_registration: bool = False
    registration_channel: str = ''      # '' = first channel in sorted order
    registration_z_level: int = 0
    dynamic_registration: bool = False  # stored and ignored, like the reference (stitcher.py:92)
    scan_pattern: str = SCAN_PATTERNS[0]
    merge_timepoints: bool = False
    merge_hcs_regions: bool = False

    def __post_init__(self) -> None:
        self.input_folder = os.path.abspath(self.input_folder)
        if self.registration_channel is None:   # argparse leaves None when the flag is not given
            self.registration_channel = ''

    def validate(self) -> None:
        """ValueError for a missing folder, an unknown format or pattern, a negative z
        (reference stitcher_parameters.py:36-59)."""
        problems = (
            (not os.path.exists(self.input_folder), f"Input folder does not exist: {self.input_folder}"),
            (self.output_format not in OUTPUT_FORMATS, "Output format must be either .ome.zarr or .ome.tiff"),
            (self.scan_pattern not in SCAN_PATTERNS, "Scan pattern must be either 'Unidirectional' or 'S-Pattern'"),
            (self.use_registration and self.registration_z_level < 0, "Registration Z-level must be non-negative"),
        )
        for failed, message in problems:
            if failed:
                raise ValueError(message)

    @property
    def stitched_folder(self) -> str:
        """``<input>_stitched_<timestamp>``: a new output tree per run (reference :61-64)."""
        return f"{self.input_folder}_stitched_{datetime.now():%Y-%m-%d_%H-%M-%S.%f}"

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> 'StitchingParameters':
        """Unknown keys are ignored, like the reference does."""
        known = {f.name for f in dataclasses.fields(cls)}
        return cls(**{key: value for key, value in data.items() if key in known})

    @classmethod
    def from_json(cls, json_path: str) -> 'StitchingParameters':
        with open(json_path) as fh:
            return cls.from_dict(json.load(fh))

    def to_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)

    def to_json(self, json_path: str) -> None:
        with open(json_path, 'w') as fh:
            json.dump(self.to_dict(), fh, indent=2)
