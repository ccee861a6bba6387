#pragma once
// declaration-only stand-in (see README.md): the fields of geometry_msgs/TransformStamped the binding reads
namespace geometry_msgs {
struct Vector3 { double x, y, z; };
struct Quaternion { double x, y, z, w; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { Transform transform; };
}
