// ndlstream.h -- the `field=value` text-stream conventions of GPc's model files (reference CNdlInterfaces.h:21-200,
// ndlstrutil.cpp:26-36): one field per line, lines starting with '#' are comments, numbers are read with atof / strtod
// (so the hexadecimal floats newer libstdc++ builds of the reference emit -- "0x1.999999999999ap-3" -- read as well as
// plain decimals).  Header-only glue for the readers in CMatrix / CKern / CNoise / CGp.
#ifndef GPC_AMD_NDLSTREAM_H
#define GPC_AMD_NDLSTREAM_H
#include <cstdlib>
#include <istream>
#include <string>
#include "ndlexceptions.h"

namespace ndlstream {
// ndlstrutil::getline: next line that is not a comment; '\r' stripped
inline bool getline(std::istream& in, std::string& line)
{
  bool got = false;
  do {
    got = static_cast<bool>(std::getline(in, line));
    if(got && !line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
  } while(got && !line.empty() && line[0] == '#');
  return got;
}
// CStreamInterface::readStringFromStream (CNdlInterfaces.h:88-97)
inline std::string readField(std::istream& in, const std::string& name)
{
  std::string line;
  if(!getline(in, line)) throw ndlexceptions::StreamFormatError(name, "unexpected end of stream");
  const size_t eq = line.find('=');
  if(eq == std::string::npos || line.substr(0, eq) != name) throw ndlexceptions::StreamFormatError(name, "got '" + line + "'");
  return line.substr(eq + 1);
}
inline long readInt(std::istream& in, const std::string& name) { return std::atol(readField(in, name).c_str()); }
inline double readDouble(std::istream& in, const std::string& name) { return std::strtod(readField(in, name).c_str(), 0); }
inline bool readBool(std::istream& in, const std::string& name) { return readInt(in, name) != 0; }
// CStreamInterface::readVersionFromStream (CNdlInterfaces.h:37-44): MINVERSION = 0.2
inline double readVersion(std::istream& in)
{
  const double ver = readDouble(in, "version");
  if(ver < 0.2 - 1e-12) throw ndlexceptions::StreamFormatError("version", "stream written by an incompatible version");
  return ver;
}
}  // namespace ndlstream
#endif
