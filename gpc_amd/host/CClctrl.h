// CClctrl.h -- command-line plumbing of the `gp` tool (reference CClctrl.h:20-160, CClctrl.cpp): argv walking,
// verbosity / seed / file-format flags and the SVMlight reader.  Host-only glue.
#ifndef GPC_AMD_CCLCTRL_H
#define GPC_AMD_CCLCTRL_H
#include <string>
#include "CMatrix.h"
#include "ndlutil.h"

class CClctrl {
 public:
  CClctrl(int argc, char** argv);
  virtual ~CClctrl() {}
  // argument walking
  bool isFlags() const { return flags; }
  void setFlags(bool v) { flags = v; }
  bool isCurrentArgumentFlag() const;
  bool isCurrentArg(const std::string& shortName, const std::string& longName) const;
  std::string getCurrentArgument() const;
  int getCurrentArgumentNo() const { return argNo; }
  void incrementArgument() { argNo++; }
  int getIntFromCurrentArgument() const;
  double getDoubleFromCurrentArgument() const;
  bool getBoolFromCurrentArgument() const;
  void unrecognisedFlag();
  void exitError(const std::string& error);   // never blocks on stdin (the reference's cin.get(), CClctrl.cpp:49-54, is dropped)
  void exitNormal();
  // settings
  int getVerbosity() const { return verbosity; }
  void setVerbosity(int v) { verbosity = v; }
  unsigned long getSeed() const { return seed; }
  void setSeed(unsigned long s) { seed = s; ndlutil::init_genrand(s); }   // CClctrl.h:78-81
  int getFileFormat() const { return fileFormat; }
  void setFileFormat(int f) { fileFormat = f; }
  std::string getMode() const { return mode; }
  void setMode(const std::string& m) { mode = m; }
  // data
  void readData(CMatrix& X, CMatrix& y, const std::string fileName);
  void readSvmlDataFile(CMatrix& X, CMatrix& y, const std::string fileName);   // CClctrl.cpp:57-180
  virtual void helpInfo() {}

 protected:
  int argc;
  char** argv;

 private:
  int argNo;
  bool flags;
  int verbosity;
  unsigned long seed;
  int fileFormat;
  std::string mode;
};
#endif
