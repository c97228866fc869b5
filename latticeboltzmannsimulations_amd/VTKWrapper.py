"""saveToVTK -- the output boundary of the reference (VTKWrapper.py:6-10), same signature and
byte-identical files, written directly (no pyevtk): a VTK XML RectilinearGrid ``.vtr`` with
raw appended binary blocks (UInt64 block headers, little endian, x fastest), cell arrays
``velocity`` (3 components interleaved) and ``pressure`` (= rho).

tests/test_host_logic.py compares the bytes with files produced by the reference's own writer
(tests/golden/vtr_*.npz).  Like the reference, X*Y cell values are attached to the grid whose
points are ``grid`` (so the extent is 0..X-1, 0..Y-1, 0..0) -- kept as is for drop-in parity: that
grid has only (X-1)*(Y-1) cells (MRT.py:91-94,606-610 vs pyevtk/hl.py:152-153), so readers see a
size mismatch.  ``correct=True`` writes the same arrays as POINT data of the same grid -- one value
per lattice node, which is what the X*Y values are -- and is the mode to use for new output.
"""
import numpy as np

_VTK_TYPE = {np.dtype("float64"): "Float64", np.dtype("float32"): "Float32",
             np.dtype("int32"): "Int32", np.dtype("int64"): "Int64"}


def _tname(a):
    try:
        return _VTK_TYPE[np.asarray(a).dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype for VTK output: {np.asarray(a).dtype}")


def saveToVTK(velocity, rho, prefix, saveNumber, grid, correct=False):
    """velocity: (ux, uy, uz) arrays of shape [X, Y, 1]; rho: [X, Y, 1]; grid: (x, y, z)
    coordinate vectors.  Writes ./{prefix}.{saveNumber}.vtr and returns its path.
    correct=False: byte-identical to the reference's writer (cell data, see the module text);
    correct=True: the values as point data of the same grid (a consistent file)."""
    name = "./" + prefix + "." + saveNumber + ".vtr"
    x, y, z = (np.ascontiguousarray(g) for g in grid)
    vx, vy, vz = (np.asarray(v) for v in velocity)
    rho = np.asarray(rho)
    if not (vx.shape == vy.shape == vz.shape == rho.shape) or vx.ndim != 3:
        raise ValueError("velocity components and rho must share one [X, Y, Z] shape")
    if not (vx.dtype == vy.dtype == vz.dtype):
        raise ValueError("velocity components must share one dtype")
    ncells = vx.size
    if correct and vx.shape != (x.size, y.size, z.size):
        raise ValueError("correct=True: one value per grid point, i.e. arrays of shape (len(x), len(y), len(z))")
    end = (x.size - 1, y.size - 1, z.size - 1)
    ext = "%d %d %d %d %d %d" % (0, end[0], 0, end[1], 0, end[2])

    blocks = [(x, 1), (y, 1), (z, 1)]
    offs, off = [], 0
    for a, nc in blocks:
        offs.append(off)
        off += 8 + a.size * a.dtype.itemsize
    off_vel = off
    off += 8 + 3 * ncells * vx.dtype.itemsize
    off_p = off

    hdr = []
    hdr.append('<?xml version="1.0"?>\n')
    hdr.append('<VTKFile type="RectilinearGrid" version="1.0" byte_order="LittleEndian" header_type="UInt64">\n')
    hdr.append('<RectilinearGrid WholeExtent="%s">\n' % ext)
    hdr.append('<Piece Extent="%s">\n' % ext)
    hdr.append('<Coordinates>\n')
    for nm, a, o in (("x_coordinates", x, offs[0]), ("y_coordinates", y, offs[1]), ("z_coordinates", z, offs[2])):
        hdr.append('<DataArray Name="%s" NumberOfComponents="1" type="%s" format="appended" offset="%d"/>\n'
                   % (nm, _tname(a), o))
    hdr.append('</Coordinates>\n')
    section = "PointData" if correct else "CellData"
    hdr.append('<%s scalars="velocity">\n' % section)
    hdr.append('<DataArray Name="velocity" NumberOfComponents="3" type="%s" format="appended" offset="%d"/>\n'
               % (_tname(vx), off_vel))
    hdr.append('<DataArray Name="pressure" NumberOfComponents="1" type="%s" format="appended" offset="%d"/>\n'
               % (_tname(rho), off_p))
    hdr.append('</%s>\n' % section)
    hdr.append('</Piece>\n')
    hdr.append('</RectilinearGrid>\n')
    hdr.append('<AppendedData encoding="raw">\n_')

    le = "<"
    with open(name, "wb") as f:
        f.write("".join(hdr).encode("ascii"))
        for a, _ in blocks:
            f.write(np.array([a.size * a.dtype.itemsize], dtype=le + "u8").tobytes())
            f.write(a.astype(a.dtype.newbyteorder(le), copy=False).tobytes())
        f.write(np.array([3 * ncells * vx.dtype.itemsize], dtype=le + "u8").tobytes())
        inter = np.empty((ncells, 3), dtype=vx.dtype.newbyteorder(le))
        inter[:, 0] = vx.ravel(order="F")
        inter[:, 1] = vy.ravel(order="F")
        inter[:, 2] = vz.ravel(order="F")
        f.write(inter.tobytes())
        f.write(np.array([ncells * rho.dtype.itemsize], dtype=le + "u8").tobytes())
        f.write(rho.astype(rho.dtype.newbyteorder(le), copy=False).ravel(order="F").tobytes())
        f.write(b"\n</AppendedData>\n</VTKFile>")
    return name
