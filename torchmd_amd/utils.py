"""Observability / configuration glue with the reference module's names (`torchmd/utils.py`): the CSV
monitor writer, the argument-file loader and saver of the driver, and the `.npy` -> `.xyz` converter.
Nothing here touches the device."""

from __future__ import annotations

import argparse
import csv
import json
import os
import time

import numpy as np
import yaml


class LogWriter:
    """CSV monitor (`utils.py:10-38`): one row per output period, column `t` = seconds since creation;
    `header` (str or dict) goes on a comment line above the column names."""

    def __init__(self, path, keys, header="", name="monitor.csv"):
        if path is None:
            raise ValueError("LogWriter needs a directory")
        self.keys = tuple(keys) + ("t",)
        os.makedirs(path, exist_ok=True)
        filename = os.path.join(path, name)
        self.f = open(filename, "wt")  # truncates an older monitor of the same name
        if isinstance(header, dict):
            header = "# {} \n".format(json.dumps(header))
        self.f.write(header)
        self.logger = csv.DictWriter(self.f, fieldnames=self.keys)
        self.logger.writeheader()
        self.f.flush()
        self.tstart = time.time()

    def write_row(self, epinfo):
        row = dict(epinfo, t=time.time() - self.tstart)
        self.logger.writerow(row)
        self.f.flush()


class LoadFromFile(argparse.Action):
    """`parser.add_argument("--conf", type=open, action=LoadFromFile)`: a YAML file updates the namespace
    key by key; any other file is read as `key=value` lines cast to the type of the option's default."""

    def __call__(self, parser, namespace, values, option_string=None):
        with values as fh:
            if values.name.endswith((".yaml", ".yml")):
                namespace.__dict__.update(yaml.safe_load(fh) or {})
                return
            for line in fh.read().rstrip().split("\n"):
                if not line.strip():
                    continue
                key, value = line.split("=", 1)
                current = namespace.__dict__.get(key)
                namespace.__dict__[key] = type(current)(value) if current is not None else value


def save_argparse(args, filename, exclude=None):
    """Write the options back to a `.yaml` (or `key=value`) file, leaving out `exclude`."""
    skip = {exclude} if isinstance(exclude, str) else set(exclude or ())
    items = {k: v for k, v in vars(args).items() if k not in skip}
    with open(filename, "w") as fh:
        if filename.endswith((".yaml", ".yml")):
            yaml.safe_dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in items.items()}, fh)
        else:
            for k, v in items.items():
                fh.write(f"{k}={v}\n")


def xyz_writer(input_file, output_file, mol_elements):
    """Append the frames of an `[N,3,F]` `.npy` trajectory to an `.xyz` file."""
    traj = np.load(input_file)
    natoms, _, nframes = traj.shape
    with open(output_file, "a") as fh:
        for frame in range(nframes):
            fh.write(f"{natoms}\n\n")
            for atom in range(natoms):
                fh.write(f"{mol_elements[atom]} " + " ".join(map(str, traj[atom, :, frame])) + "\n")
