"""Per-tile trainer used by the launcher test: writes a marker in the tile's checkpoint dir and reports 7 iterations."""
import os


def train_tile(tile_dir, out_paths, device, tile_index):
    with open(os.path.join(out_paths["chkpnt"], "done.txt"), "w") as f:
        f.write(f"{os.path.basename(tile_dir)} {tile_index} {device.type} rank{os.environ.get('RANK', '0')}\n")
    return 7


def train_tile_rank1_dies(tile_dir, out_paths, device, tile_index):
    import os
    if os.environ.get("RANK") == "1":
        raise SystemExit(5)
    return 1
