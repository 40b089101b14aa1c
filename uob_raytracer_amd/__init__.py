"""MI355X-native Cornell-Box ray tracer: host-side mirror of the reference's interface over libuob_rt.so."""
