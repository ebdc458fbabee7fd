// TEST INFRASTRUCTURE ONLY: just enough of boost::serialization for the vendored DBoW2 headers to parse (the serialize()
// member templates are never instantiated by the oracle's reference build).
#pragma once
namespace boost { namespace serialization {
class access {};
template <class Base, class Derived> Base &base_object(Derived &d) { return static_cast<Base &>(d); }
}}  // namespace boost::serialization
