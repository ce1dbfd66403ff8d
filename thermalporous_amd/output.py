"""Field output for ParaView: the reference writes `results/pressure.pvd`, `temperature.pvd` and `saturation_o.pvd`
through Firedrake's `File` at t = 0 and every `n_save` time steps (thermalmodel.py:113-133, 303-322).  Here each
`File` is a PVD collection of VTK ImageData (`.vti`) pieces -- the grids are structured boxes, cell data, raw
little-endian float64 base64-inlined -- written with the standard library only."""
import base64
import os
import struct

import numpy as np


class File():
    def __init__(self, path):
        self.path = path
        self.base = os.path.splitext(path)[0]
        self.entries = []          # (time, vti file name)
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)

    def write(self, name, values, geo, time=0.0):
        """values: one double per cell, x fastest then y then z (the model's field-major order)."""
        nx, ny, nz = geo.Nx, geo.Ny, getattr(geo, "Nz", 1)
        dx, dy, dz = geo.Dx, geo.Dy, getattr(geo, "Dz", 1.0)
        a = np.ascontiguousarray(np.asarray(values, dtype="<f8").reshape(-1))
        if a.size != nx*ny*nz:
            raise ValueError("field has %d values for a %dx%dx%d grid" % (a.size, nx, ny, nz))
        raw = a.tobytes()
        payload = base64.b64encode(struct.pack("<I", len(raw)) + raw).decode()     # uint32 byte count header
        fn = "%s_%d.vti" % (self.base, len(self.entries))
        with open(fn, "w") as f:
            f.write('<?xml version="1.0"?>\n<VTKFile type="ImageData" version="0.1" byte_order="LittleEndian">\n')
            f.write('  <ImageData WholeExtent="0 %d 0 %d 0 %d" Origin="0 0 0" Spacing="%r %r %r">\n'
                    % (nx, ny, nz, dx, dy, dz))
            f.write('    <Piece Extent="0 %d 0 %d 0 %d">\n      <CellData Scalars="%s">\n' % (nx, ny, nz, name))
            f.write('        <DataArray type="Float64" Name="%s" format="binary">%s</DataArray>\n' % (name, payload))
            f.write('      </CellData>\n    </Piece>\n  </ImageData>\n</VTKFile>\n')
        self.entries.append((float(time), os.path.basename(fn)))
        with open(self.path, "w") as f:
            f.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1" byte_order="LittleEndian">\n')
            f.write('  <Collection>\n')
            for t, e in self.entries:
                f.write('    <DataSet timestep="%r" part="0" file="%s"/>\n' % (t, e))
            f.write('  </Collection>\n</VTKFile>\n')


def read_vti(path):
    """(name, values) of the single cell array of a file written above (used by the tests)."""
    import xml.etree.ElementTree as ET
    arr = ET.parse(path).getroot().find("ImageData/Piece/CellData/DataArray")
    blob = base64.b64decode(arr.text.strip())
    n = struct.unpack("<I", blob[:4])[0]
    return arr.get("Name"), np.frombuffer(blob[4:4 + n], dtype="<f8")
