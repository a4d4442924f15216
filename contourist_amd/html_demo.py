"""three.js emitters for an extracted surface -- the consumer step right after the march in the reference
(`contourist/html_demo.py`: emit_three_json :147-161, grid_html_page :118-131).  Pure host-side string
formatting of (points, triangles); accepts any object with get_points_and_triangles() (the device-backed
TriangulatedIsosurfaces / GridContour3d) or a (points, triangles) tuple.
"""
import json



def _mesh(source):
    if hasattr(source, "get_points_and_triangles"):
        return source.get_points_and_triangles()
    return source


# the three.js "Geometry" JSON (format 3) as the reference writes it, byte for byte (html_demo.py:133-161): one number per
# line inside the two arrays, every face introduced by its type code 0 (plain triangle)
THREE_JSON = """
{
    "metadata": {
        "version": 3,
        "type": "Geometry",
        "generator": "GeometryExporter"
    },
    "faces": [%s],
    "vertices": [%s],
    "normals": [],
    "uvs": []
}
"""


def emit_three_json(grid_contour):
    "three.js JSON geometry (format 3): faces as 0, a, b, c runs, vertices flattened; the reference's exact text"
    (points, triangles) = _mesh(grid_contour)
    faces = ",\n".join("0,\n" + ",\n".join(str(int(index)) for index in triangle) for triangle in triangles)
    vertices = ",\n".join(str(coordinate) for point in points for coordinate in point)
    return THREE_JSON % (faces, vertices)


PAGE = """<!DOCTYPE html>
<html><head><title>%(title)s</title><style>body{margin:0;overflow:hidden}</style></head>
<body><div id="%(target_div)s"></div>
<script src="%(load_three)s"></script>
<script>
var vertices = %(vertices)s, indices = %(indices)s;
var scene = new THREE.Scene();
var camera = new THREE.PerspectiveCamera(45, window.innerWidth / window.innerHeight, 0.1, 1000);
camera.position.set(%(camera_x)s, %(camera_y)s, %(camera_z)s); camera.lookAt(scene.position);
var renderer = new THREE.WebGLRenderer(); renderer.setSize(window.innerWidth, window.innerHeight);
document.getElementById("%(target_div)s").appendChild(renderer.domElement);
var geom = new THREE.Geometry();
vertices.forEach(function (v) { geom.vertices.push(new THREE.Vector3(v[0], v[1], v[2])); });
indices.forEach(function (t) { geom.faces.push(new THREE.Face3(t[0], t[1], t[2])); });
geom.computeFaceNormals();
var material = new THREE.MeshNormalMaterial(); material.side = THREE.DoubleSide;
scene.add(new THREE.Mesh(geom, material));
renderer.render(scene, camera);
</script></body></html>
"""


def grid_html_page(gridcontour, title="3d contour", load_three="https://cdnjs.cloudflare.com/ajax/libs/three.js/r79/three.min.js",
                   x=-30, y=40, z=50):
    "stand-alone HTML page that renders the surface with three.js"
    (points, triangles) = _mesh(gridcontour)
    return PAGE % {
        "title": title, "target_div": "THREE_OUTPUT", "load_three": load_three,
        "vertices": json.dumps([[float(c) for c in p] for p in points]),
        "indices": json.dumps([[int(i) for i in t] for t in triangles]),
        "camera_x": x, "camera_y": y, "camera_z": z,
    }
