#pragma once
// declaration-only stand-in (see README.md): nothing of pcl_conversions is named by the class header or the binding
